// encoder.1's forward - Conv2d(32 -> 64, 3x3, stride 2, pad 1) on LeakyReLU(BN(y0)) with the BatchNorm statistics of its output - as a
// ROW-STREAMING kernel for 64-pixel-wide inputs (128x128 images) and 16-bit storage, gfx950.  The tiled kernel (down2_kernel) takes
// 43 us for 101 MB; same recipe as upfinal_stream.cuh / convout_stream.cuh:
//
//   * one 768-thread workgroup per CU walks the rows of an image (or band): y0 arrives by LDS-DMA FOUR ticks ahead (a tick is
//     only ~1.5k cycles of work here: with two ticks of flight time the tick took as long as the copy's latency, 3.9k cycles),
//     never through registers;
//   * group B (waves 4..11): copies, and the BatchNorm + LeakyReLU map of 4 input rows per tick from the raw ring into the a ring.
//     Both rings keep a row as two planes - even pixels, odd pixels (one zero pad pixel in front of the odd plane: the left
//     border) - so that the stride-2 taps of 32 consecutive outputs are 32 consecutive cells of a plane (conflict-free b128
//     reads with the usual XOR swizzle); the LDS-DMA fills that layout directly (the permutation is on its source side);
//   * group A (waves 0..3): one (output row, 32-channel half) each per tick, transposed MFMAs (pixels = N, channels = M), the
//     whole 3x3x32 weight slice of the half in REGISTERS (72 VGPRs: only the 18 pixel fragments of a tick come from LDS), two
//     accumulators (even / odd k-steps), then bias, rounding, statistics of the rounded values, a wave-private LDS tile in output
//     order, 64-byte runs per pixel to memory;
//   * one raw s_barrier per tick (2 output rows): group A works on rows staged in earlier ticks.
#pragma once
#include "conv_mfma.cuh"
#include "conv_deep.cuh"
#include "convout_stream.cuh"

template <typename T> struct DnFirstStreamArgs {
    const T* yin; const float* coef; float slope; BnFuse fuse;   // y0 [B,64,64,32] and its BatchNorm (batch statistics or coefficient block)
    const T* wp; const float* bias;                              // packed [9][4][64][8] (tap = 3*ky + kx, K = input channel), bias [64]
    T* out; double* stat;                                        // y1 [B,32,32,64]; [rep][2][64] sum y | sum y^2
    int B, RB, nb, n_units;                                      // RB output rows per band, nb bands per image
};

namespace dfs {
static constexpr int WI = 64, HI = 64, WO = 32, HO = 32, NRING = 12, DD = 4, NYR = 4 * (DD + 1), P1 = 2048, ROW = P1 + 33 * 64, OPITCH = 72, OTILE = 32 * OPITCH;
// byte offset inside a ring row of chunk c of the cell at plane position j (plane 1: position 0 is the pad, pixel 2j-1 sits at j)
__device__ __forceinline__ int cell(int plane, int j, int c) { return plane * P1 + j * 64 + ((c ^ ((j >> 2) & 3)) << 4); }
}
static inline size_t dnfirst_stream_lds() { return (size_t)(dfs::NRING + dfs::NYR) * dfs::ROW + 4 * dfs::OTILE + (32 + 32 + 64) * 4 + 4 * 128 * 4; }

template <typename T>
__global__ __launch_bounds__(768) void dnfirst_stream_kernel(DnFirstStreamArgs<T> a) {
    using namespace dfs;
    typedef typename H16<T>::v8 T8;
    typedef __attribute__((ext_vector_type(4))) T T4;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* yring = smem;                                   // raw y0 rows (plane layout), filled by LDS-DMA
    char* aring = yring + NYR * ROW;                      // LeakyReLU(BN(y0)), same layout; the pad cell of every row stays zero
    char* otile0 = aring + NRING * ROW;                   // [4 waves][32 pixels][72 B]
    float* cf = reinterpret_cast<float*>(otile0 + 4 * OTILE);   // scale[32] | shift[32] | bias[64]
    float* red = cf + 128;                                // [4][128]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), r = lane & 31, h = lane >> 5;
    const int G = gridDim.x, K = a.RB / 2 + 2;

    if (tid < 32) {
        if (a.fuse.mode != BNF_NONE) { float k1; bn_fused_channel(a.fuse, tid, blockIdx.x == 0, cf[tid], k1, cf[32 + tid]); }
        else { cf[tid] = a.coef[tid]; cf[32 + tid] = a.coef[2 * 32 + tid]; }
    }
    if (tid >= 64 && tid < 128) cf[tid] = a.bias[tid - 64];
    for (int i = tid; i < NRING * ROW / 16; i += 768) *reinterpret_cast<f32x4*>(aring + i * 16) = f32x4{0.f, 0.f, 0.f, 0.f};

    if (wave >= 4) {
        // ====================================== group B: copies and staging ======================================
        const int wq = wave - 4, pt = tid - 256;
        // LDS-DMA of one tick: input rows sB .. sB+3 into y-ring slots ya .. ya+3 (wave: row wq >> 1, pieces 2 (wq & 1), +1 of the row's
        // four 1 KiB pieces: plane 0 positions 0..63 / 64..127, plane 1 positions 1..16 / 17..32 x 4 chunks).
        const int drow = wq >> 1;
        int dsrc[2], ddst[2];                  // per piece: source byte offset in the global row, destination byte offset in the ring row
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
            const int pz = 2 * (wq & 1) + jj;
            const int plane = pz >> 1, pos = (pz & 1) * 64 + lane + (plane ? 4 : 0);      // chunk position inside the plane
            const int j = pos >> 2, cs = pos & 3, px = plane ? 2 * j - 1 : 2 * j, ch = cs ^ ((j >> 2) & 3);
            dsrc[jj] = px * 64 + ch * 16;
            ddst[jj] = plane * P1 + ((pz & 1) * 64 + (plane ? 4 : 0)) * 16;                 // (wave-uniform: the copy adds lane * 16)
        }
        int ua = blockIdx.x, ka = 0, ya = 0, ba = 0, r0a = 0;
        if (ua < a.n_units) { ba = ua / a.nb; r0a = (ua - ba * a.nb) * a.RB; }
        auto issue_ahead = [&]() __attribute__((always_inline)) {
            const bool live = ua < a.n_units;
            if (live) {
                const int row = 2 * r0a - 1 + 4 * ka + drow, rlast = 2 * (r0a + a.RB) - 1;
                const bool ok = row >= 0 && row < HI && row >= 2 * r0a - 1 && row <= rlast;
                const char* rowp = reinterpret_cast<const char*>(a.yin + ((size_t)(ba * HI + (ok ? row : 0)) * WI) * 32);
                int slot = ya + drow; slot = slot >= NYR ? slot - NYR : slot;
                char* dst = yring + slot * ROW;
                cos::dma16(rowp + dsrc[0], dst + ddst[0]);
                cos::dma16(rowp + dsrc[1], dst + ddst[1]);
            }
            ya = ya + 4 >= NYR ? ya + 4 - NYR : ya + 4;
            if (++ka == K) {
                ka = 0; ua += G;
                if (ua < a.n_units) { ba = ua / a.nb; r0a = (ua - ba * a.nb) * a.RB; }
            }
            return live ? 2 : 0;
        };
        // copies of ticks 0 .. DD-1; n1..n3: how many copies the three youngest issues put in flight (the wait before a tick's closing
        // barrier lets exactly those stay outstanding: the copies of the NEXT tick are older than them)
        issue_ahead();
        int n1 = issue_ahead(), n2 = issue_ahead(), n3 = issue_ahead();
        deep::barrier_lds();                 // cf, zeroed a ring published
        // staging: chunks pt and pt + 512 of the tick's 4 x 256 (row = chunk >> 8): same cell of rows (0,1) -> u = 0, (2,3) -> u = 1
        const int srow = pt >> 8, sc_ = pt & 255, splane = sc_ >> 7, spos = (sc_ & 127) + (splane ? 4 : 0);
        const int soff = splane * P1 + spos * 16;
        f32x2 kc[4], kh[4];
        {
            const int j = spos >> 2, ch = (spos & 3) ^ ((j >> 2) & 3);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                kc[e] = f32x2{cf[ch * 8 + 2 * e], cf[ch * 8 + 2 * e + 1]};
                kh[e] = f32x2{cf[32 + ch * 8 + 2 * e], cf[32 + ch * 8 + 2 * e + 1]};
            }
        }
        cos::wait_vm(n1 + n2 + n3);          // tick 0's copies landed
        deep::barrier_lds();
        int py = 0, pyy = 0;                 // a-ring / y-ring slot of row sB
        for (int unit = blockIdx.x; unit < a.n_units; unit += G) {
            const int r0 = (unit % a.nb) * a.RB, rlast = 2 * (r0 + a.RB) - 1;
            for (int k = 0; k < K; ++k) {
                n1 = n2; n2 = n3; n3 = issue_ahead();      // copies of tick k + DD
                const int sB = 2 * r0 - 1 + 4 * k;
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int row = sB + srow + 2 * u;
                    const bool ok = row >= 0 && row < HI && row <= rlast;
                    int slot = py + srow + 2 * u; slot = slot >= NRING ? slot - NRING : slot;
                    int yslot = pyy + srow + 2 * u; yslot = yslot >= NYR ? yslot - NYR : yslot;
                    char* adst = aring + slot * ROW + soff;
                    if (!ok) { *reinterpret_cast<T8*>(adst) = T8{0, 0, 0, 0, 0, 0, 0, 0}; continue; }   // (wave-uniform) outside the image / band: a = 0
                    const T8 yv = *reinterpret_cast<const T8*>(yring + yslot * ROW + soff);
                    T8 o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        f32x2 z = f32x2{(float)yv[2 * e], (float)yv[2 * e + 1]} * kc[e] + kh[e];
                        const f32x2 zs = z * a.slope;
                        z.x = fmaxf(z.x, zs.x); z.y = fmaxf(z.y, zs.y);
                        o[2 * e] = (T)z.x; o[2 * e + 1] = (T)z.y;
                    }
                    *reinterpret_cast<T8*>(adst) = o;
                }
                cos::wait_vm(n1 + n2 + n3);  // the next tick's copies landed (the three younger issues may be outstanding)
                deep::barrier_lds();
                py = py + 4 >= NRING ? py + 4 - NRING : py + 4; pyy = pyy + 4 >= NYR ? pyy + 4 - NYR : pyy + 4;
            }
        }
    } else {
        // ====================================== group A: MFMAs, epilogue, stores ======================================
        const int arow = wave >> 1, mt = wave & 1;                 // this wave: output row oA = r0 + 2 (k - 2) + arow, channels 32 mt ..
        char* otile = otile0 + wave * OTILE;
        // the half's weights: A[m = channel 32 mt + r][k = input channel] of tap t, k-step ks
        Frag<T> wreg[9][2];
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) wreg[t][ks] = load_frag(a.wp + ((size_t)((t * 4 + 2 * ks + h) * 64 + 32 * mt + r)) * 8);
        // pixel fragments: a[input pixel 2 r + kx - 1][channels 16 ks + 8 h ..]: kx = 0 -> odd plane position r, 1 -> even plane r, 2 -> odd plane r + 1
        int offB[3][2];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) { offB[0][ks] = cell(1, r, 2 * ks + h); offB[1][ks] = cell(0, r, 2 * ks + h); offB[2][ks] = cell(1, r + 1, 2 * ks + h); }
        deep::barrier_lds();
        const float* biap = cf + 64 + 32 * mt + 4 * h;             // bias of the lane's channel pairs: 32 mt + 4h + 2 (e & 1) + 8 (e >> 1)
        f32x2 s1[8], s2[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) { s1[e] = f32x2{0.f, 0.f}; s2[e] = f32x2{0.f, 0.f}; }
        deep::barrier_lds();
        int pa = 0;                          // ring slot of input row sB of the current tick; this tick's rows start 8 slots back
        for (int unit = blockIdx.x; unit < a.n_units; unit += G) {
            const int b = unit / a.nb, r0 = (unit - b * a.nb) * a.RB, r1 = r0 + a.RB;
            for (int k = 0; k < K; ++k) {
                const int oA = r0 + 2 * (k - 2) + arow;
                if (oA >= r0 && oA < r1) {
                    f32x16 acc0, acc1;
#pragma unroll
                    for (int i = 0; i < 16; ++i) { acc0[i] = 0.f; acc1[i] = 0.f; }
#pragma unroll
                    for (int ky = 0; ky < 3; ++ky) {
                        int slot = pa + NRING - 8 + 2 * arow + ky; slot = slot >= NRING ? slot - NRING : slot; slot = slot >= NRING ? slot - NRING : slot;
                        const char* row = aring + slot * ROW;      // input row 2 oA - 1 + ky
#pragma unroll
                        for (int kx = 0; kx < 3; ++kx) {
                            const Frag<T> b0 = load_frag(reinterpret_cast<const T*>(row + offB[kx][0]));
                            const Frag<T> b1 = load_frag(reinterpret_cast<const T*>(row + offB[kx][1]));
                            mma(acc0, wreg[ky * 3 + kx][0], b0);
                            mma(acc1, wreg[ky * 3 + kx][1], b1);
                        }
                    }
                    // epilogue: bias, round to storage, statistics of the rounded values, pixel-major tile
                    f32x2 bia[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) bia[e] = *reinterpret_cast<const f32x2*>(biap + 2 * (e & 1) + 8 * (e >> 1));
                    char* cellp = otile + r * OPITCH + 8 * h;
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        T4 o4;
#pragma unroll
                        for (int e2 = 0; e2 < 2; ++e2) {
                            const int e = 2 * g + e2;
                            const T o0 = (T)(acc0[2 * e] + acc1[2 * e] + bia[e].x), o1 = (T)(acc0[2 * e + 1] + acc1[2 * e + 1] + bia[e].y);
                            const f32x2 v = f32x2{(float)o0, (float)o1};
                            s1[e] += v;
                            s2[e] = f32x2{__builtin_fmaf(v.x, v.x, s2[e].x), __builtin_fmaf(v.y, v.y, s2[e].y)};
                            o4[2 * e2] = o0; o4[2 * e2 + 1] = o1;
                        }
                        *reinterpret_cast<T4*>(cellp + g * 16) = o4;
                    }
                    // (wave-private tile, LDS executes the wave's accesses in order) per pixel the half's 64 B of the 128 B output pixel
                    char* dg = reinterpret_cast<char*>(a.out + ((size_t)(b * HO + oA) * WO) * 64) + 64 * mt;
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int id = lane + 64 * u, px = id >> 3, pc = id & 7;
                        *reinterpret_cast<T4*>(dg + px * 128 + pc * 8) = *reinterpret_cast<const T4*>(otile + px * OPITCH + pc * 8);
                    }
                }
                deep::barrier_lds();
                pa = pa + 4 >= NRING ? pa + 4 - NRING : pa + 4;
            }
        }
        // statistics: lanes r of a half-wave hold the same 16 channels
#pragma unroll
        for (int e = 0; e < 8; ++e) {
#pragma unroll
            for (int o = 16; o > 0; o >>= 1) {
                s1[e].x += __shfl_xor(s1[e].x, o, 64); s1[e].y += __shfl_xor(s1[e].y, o, 64);
                s2[e].x += __shfl_xor(s2[e].x, o, 64); s2[e].y += __shfl_xor(s2[e].y, o, 64);
            }
        }
        if (r == 0) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int c = 32 * mt + acc_row(2 * e, lane);
                red[wave * 128 + c] = s1[e].x; red[wave * 128 + c + 1] = s1[e].y;
                red[wave * 128 + 64 + c] = s2[e].x; red[wave * 128 + 64 + c + 1] = s2[e].y;
            }
        }
    }
    deep::barrier_lds();
    if (tid < 128 && a.stat) {
        // channel c = tid & 63 belongs to the waves with mt = c >> 5: waves mt and mt + 2
        const int c = tid & 63, mt = c >> 5;
        const float v = red[mt * 128 + tid] + red[(mt + 2) * 128 + tid];
        unsafeAtomicAdd(&a.stat[stat_rep() * 128 + tid], (double)v);
    }
}
