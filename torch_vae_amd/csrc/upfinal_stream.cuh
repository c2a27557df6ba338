// final_layer.0 of the decoder - ConvTranspose2d(32 -> 32, 3x3, stride 2, pad 1, output_padding 1) on LeakyReLU(BN(y6)), forward,
// with the BatchNorm statistics of its output - as a ROW-STREAMING kernel for 64-pixel-wide inputs (128x128 images) and 16-bit
// storage, gfx950.  The layer writes 4x the pixels it reads (336 MB per step at the bench workload, 80 % of it stores): the
// tiled kernel (up2_kernel, conv_pipe.cuh) runs it at 3.5 TB/s, its phases summing at two waves per SIMD.  Same recipe as the
// output conv's streaming kernel (convout_stream.cuh):
//
//   * one 1024-thread workgroup per CU walks the rows of an image (or of a band of rows) top to bottom; y6 arrives by LDS-DMA
//     two ticks ahead into a raw ring, never through registers;
//   * group B (waves 8..15): copies, and the BatchNorm + LeakyReLU map of 4 input rows per tick from the raw ring into the
//     a ring (16-byte chunks XOR-swizzled by (pixel >> 2) & 3, one zero pad pixel at the right edge);
//   * group A (waves 0..7): one block of 32 input pixels each per tick, transposed MFMAs (pixels = N, output channels = M,
//     weights = A fragments read from an LDS copy of the packed image): an input pixel (m, n) and its neighbours (m, n+1),
//     (m+1, n), (m+1, n+1) feed the 2x2 output block (2m.., 2n..) - 1 + 2 + 2 + 4 taps for the four parities, 18 MFMAs.  The
//     two output rows of the block are finished one after the other (32 accumulator registers at a time): bias, rounding to
//     storage, statistics of the ROUNDED values in packed f32 math, 8-byte writes into a wave-private LDS tile in output order,
//     read back as one contiguous 4 KiB run per output row and stored coalesced;
//   * one raw s_barrier per tick: group A works on rows staged in earlier ticks.
#pragma once
#include "conv_mfma.cuh"
#include "conv_deep.cuh"
#include "convout_stream.cuh"

template <typename T> struct UpFinalStreamArgs {
    const T* yin; const float* coef; float slope; BnFuse fuse;   // y6 [B,64,64,32] and its BatchNorm (batch statistics or coefficient block)
    const T* wp; const float* bias;                              // packed [9][4][32][8] (tap = 3*ky + kx, K = input channel), bias [32]
    T* out; double* stat;                                        // y7 [B,128,128,32]; [rep][2][32] sum y | sum y^2
    int B, RB, nb, n_units;                                      // RB input rows per band, nb bands per image
};

namespace ufs {
// geometry of an instance: CIN input channels, WL x WL input pixels (32 output channels, 2 WL x 2 WL output pixels)
template <int CIN, int WL_> struct Geo {
    static constexpr int WL = WL_, HL = WL_, NCH = CIN / 8, PXB = CIN * 2, NKS = CIN / 16, NBR = WL / 32;
    // copies run DD ticks ahead of the staging (the 64-channel instance keeps 36 KiB of weights in LDS and has room for one tick only)
    static constexpr int NRING = 12, DD = CIN == 32 ? 2 : 1, NYR = 4 * (DD + 1);
    static constexpr int YROW = WL * PXB, AROW = (WL + 1) * PXB, OPITCH = 72, OTILE = 64 * OPITCH, WBYTES = 9 * NCH * 32 * 16;
    static_assert(WL * NCH == 256, "a row is 256 16-byte chunks (64 pixels x 32 channels or 32 pixels x 64 channels)");
    // byte offset of chunk c of pixel px in a ring row: the XOR swizzle that makes 16 consecutive pixels' b128 reads conflict-free
    __device__ static __forceinline__ int off(int px, int c) { return px * PXB + ((c ^ ((px >> (NCH == 4 ? 2 : 1)) & (NCH - 1))) << 4); }
    static constexpr size_t lds() { return (size_t)NYR * YROW + (size_t)NRING * AROW + WBYTES + 8 * OTILE + (2 * CIN + 32) * 4; }
};
static constexpr int HL = 64, WL = 64;     // final_layer.0 (the launcher's names)
}
template <int CIN, int WL> static inline size_t upfinal_stream_lds() { return ufs::Geo<CIN, WL>::lds(); }

template <typename T, int CIN, int WL_>
__global__ __launch_bounds__(1024) void upfinal_stream_kernel(UpFinalStreamArgs<T> a) {
    typedef ufs::Geo<CIN, WL_> GE;
    constexpr int WL = GE::WL, HL = GE::HL, NCH = GE::NCH, PXB = GE::PXB, NKS = GE::NKS, NBR = GE::NBR;
    constexpr int NRING = GE::NRING, DD = GE::DD, NYR = GE::NYR, YROW = GE::YROW, AROW = GE::AROW, OPITCH = GE::OPITCH, OTILE = GE::OTILE, WBYTES = GE::WBYTES;
    typedef typename H16<T>::v8 T8;
    typedef typename H16<T>::v2 T2;
    typedef __attribute__((ext_vector_type(4))) T T4;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* yring = smem;                                   // raw y6 rows, filled by LDS-DMA
    char* aring = yring + NYR * YROW;                     // LeakyReLU(BN(y)), the pad pixel at the right end of every row stays zero
    char* wlds = aring + NRING * AROW;                    // packed weights
    char* otile0 = wlds + WBYTES;                         // [8 waves][64 output pixels][72 B]
    float* cf = reinterpret_cast<float*>(otile0 + 8 * OTILE);   // scale[CIN] | shift[CIN] | bias[32]
    float* red = reinterpret_cast<float*>(otile0);        // final reduction [8][64] (the tiles are dead by then)
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), r = lane & 31, h = lane >> 5;
    const int G = gridDim.x, K = a.RB / 4 + 2;
    const int wq = wave & 7;

    if (tid < CIN) {
        if (a.fuse.mode != BNF_NONE) { float k1; bn_fused_channel(a.fuse, tid, blockIdx.x == 0, cf[tid], k1, cf[CIN + tid]); }
        else { cf[tid] = a.coef[tid]; cf[CIN + tid] = a.coef[2 * CIN + tid]; }
    }
    if (tid >= 64 && tid < 96) cf[2 * CIN + tid - 64] = a.bias[tid - 64];
    for (int i = tid; i < NRING * AROW / 16; i += 1024) *reinterpret_cast<f32x4*>(aring + i * 16) = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int i = tid; i < WBYTES / 16; i += 1024) *reinterpret_cast<f32x4*>(wlds + i * 16) = *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(a.wp) + i * 16);

    if (wave >= 8) {
        // ====================================== group B: copies and staging ======================================
        // LDS-DMA of one tick: input rows sB .. sB+3 of image b into y-ring slots ya .. ya+3 (wave: row wq >> 1, half wq & 1, two 1 KiB
        // copies).  Rows the band does not need copy row 0 (never used).
        const int drow = wq >> 1, dci = (wq & 1) * 128 + lane;                 // this lane's first chunk position in the row (second: + 64)
        int ua = blockIdx.x, ka = 0, ya = 0, ba = 0, r0a = 0;
        if (ua < a.n_units) { ba = ua / a.nb; r0a = (ua - ba * a.nb) * a.RB; }
        auto issue_ahead = [&]() __attribute__((always_inline)) {
            const bool live = ua < a.n_units;
            if (live) {
                const int row = r0a - 3 + 4 * ka + drow, r1 = r0a + a.RB;
                const bool ok = row >= r0a && row < HL && row <= r1;
                const char* rowp = reinterpret_cast<const char*>(a.yin + ((size_t)(ba * HL + (ok ? row : 0)) * WL) * CIN);
                int slot = ya + drow; slot = slot >= NYR ? slot - NYR : slot;
                char* dst = yring + slot * YROW + (wq & 1) * 2048;
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int ci = dci + 64 * j, px = ci / NCH, cs = ci % NCH;             // linear position ci = (pixel, stored slot)
                    cos::dma16(rowp + GE::off(px, cs), dst + j * 1024);                    // (source chunk = slot ^ swizzle: off() is an involution on the slot)
                }
            }
            ya = ya + 4 >= NYR ? ya + 4 - NYR : ya + 4;
            if (++ka == K) {
                ka = 0; ua += G;
                if (ua < a.n_units) { ba = ua / a.nb; r0a = (ua - ba * a.nb) * a.RB; }
            }
            return live ? 2 : 0;
        };
        int nd1 = issue_ahead();             // tick 0 (and, two ticks ahead, tick 1: then its copies may stay in flight while tick 0's land)
        if (DD == 2) nd1 = issue_ahead(); else nd1 = 0;
        deep::barrier_lds();                 // cf, zeroed a ring, weights published
        // staging: chunks st and st + 512 of the tick's 4 x 256 (row = chunk >> 8: rows 0,1 and 2,3; same pixel and channels for both)
        const int st = tid & 511, srow = st >> 8, spos = (st & 255) * 16;
        f32x2 kc[4], kh[4];
        {
            const int pos = st & 255, px = pos / NCH, ch = (GE::off(px, pos % NCH) - px * PXB) >> 4;   // the channel chunk stored at this position
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                kc[e] = f32x2{cf[ch * 8 + 2 * e], cf[ch * 8 + 2 * e + 1]};
                kh[e] = f32x2{cf[CIN + ch * 8 + 2 * e], cf[CIN + ch * 8 + 2 * e + 1]};
            }
        }
        cos::wait_vm(nd1);
        deep::barrier_lds();
        int py = 0, pyy = 0;                 // a-ring / y-ring slot of row sB
        for (int unit = blockIdx.x; unit < a.n_units; unit += G) {
            const int r0 = (unit % a.nb) * a.RB, r1 = r0 + a.RB;
            for (int k = 0; k < K; ++k) {
                const int nd = issue_ahead();
                const int sB = r0 - 3 + 4 * k;
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int row = sB + srow + 2 * u;
                    const bool ok = row >= r0 && row < HL && row <= r1;
                    int slot = py + srow + 2 * u; slot = slot >= NRING ? slot - NRING : slot;
                    int yslot = pyy + srow + 2 * u; yslot = yslot >= NYR ? yslot - NYR : yslot;
                    char* adst = aring + slot * AROW + spos;
                    if (!ok) { *reinterpret_cast<T8*>(adst) = T8{0, 0, 0, 0, 0, 0, 0, 0}; continue; }   // (wave-uniform) outside the image / band: a = 0
                    const T8 yv = *reinterpret_cast<const T8*>(yring + yslot * YROW + spos);
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
                cos::wait_vm(DD == 2 ? nd : 0);   // the next tick's copies landed (two ticks ahead: only this tick's, issued after them, may be outstanding)
                deep::barrier_lds();
                py = py + 4 >= NRING ? py + 4 - NRING : py + 4; pyy = pyy + 4 >= NYR ? pyy + 4 - NYR : pyy + 4;
            }
        }
    } else {
        // ====================================== group A: MFMAs, epilogue, stores ======================================
        // this wave's work: input row sA + arow and - 64-pixel rows - the block of 32 pixels x0 .. (both output rows of it), or - 32-pixel
        // rows - the whole row and ONE of its two output rows
        const int arow = wave >> 1, x0 = NBR == 2 ? (wave & 1) * 32 : 0, pysel = wave & 1;
        char* otile = otile0 + wave * OTILE;
        // B fragments: a[pixel x0 + r + dx][channels 16 ks + 8h ..]
        int offB[2][NKS];
#pragma unroll
        for (int dx = 0; dx < 2; ++dx)
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) offB[dx][ks] = GE::off(x0 + r + dx, 2 * ks + h);
        const int offWt = h * 512 + r * 16;                        // A fragment of (tap t, k-step ks): wlds + (t * NCH + 2 ks) * 512 + offWt
        deep::barrier_lds();
        const float* biap = cf + 2 * CIN + 4 * h;                       // bias of the lane's channel pairs: acc_row(2e, lane) = 4h + 2 (e & 1) + 8 (e >> 1)
        f32x2 s1[8], s2[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) { s1[e] = f32x2{0.f, 0.f}; s2[e] = f32x2{0.f, 0.f}; }
        deep::barrier_lds();
        int pa = 0;                          // ring slot of row sB of the current tick; this tick's rows sA = sB - 5 sit 5 slots back
        for (int unit = blockIdx.x; unit < a.n_units; unit += G) {
            const int b = unit / a.nb, r0 = (unit - b * a.nb) * a.RB, r1 = r0 + a.RB;
            for (int k = 0; k < K; ++k) {
                const int m = r0 + 4 * (k - 2) + arow;             // input row of this wave's block
                if (m >= r0 && m < r1) {
                    int s0 = pa + NRING - 5 + arow; s0 = s0 >= NRING ? s0 - NRING : s0; s0 = s0 >= NRING ? s0 - NRING : s0;
                    int s1r = s0 + 1; s1r = s1r >= NRING ? s1r - NRING : s1r;
                    const char* row0 = aring + s0 * AROW;          // row m
                    const char* row1 = aring + s1r * AROW;         // row m + 1 (zeros below the image)
                    auto bfrag = [&](const char* row, int dx, int ks) __attribute__((always_inline)) {
                        return load_frag(reinterpret_cast<const T*>(row + offB[dx][ks]));
                    };
                    auto wfrag = [&](int t, int ks) __attribute__((always_inline)) {
                        return load_frag(reinterpret_cast<const T*>(wlds + (t * NCH + 2 * ks) * 512 + offWt));
                    };
                    T* orow_g = a.out + ((size_t)(b * 2 * HL + 2 * m) * (2 * WL) + 2 * x0) * 32;   // output row 2m, pixels 2 x0 ..
#pragma unroll
                    for (int py_ = 0; py_ < 2; ++py_) {
                        if (NBR == 1 && py_ != pysel) continue;     // (wave-uniform)
                        f32x16 acc[2];                              // output parity (py_, 0), (py_, 1)
#pragma unroll
                        for (int q = 0; q < 2; ++q)
#pragma unroll
                            for (int i = 0; i < 16; ++i) acc[q][i] = 0.f;
#pragma unroll
                        for (int ks = 0; ks < NKS; ++ks) {
                            // (m, n), (m, n+1) and - odd output rows - (m+1, n), (m+1, n+1) of this k-step; the fence keeps the other
                            // k-step's fragments out of the register file (the wave has 128 VGPRs)
                            const Frag<T> bm0 = bfrag(row0, 0, ks), bm1 = bfrag(row0, 1, ks);
                            if (py_ == 0) {     // oy = 2m: ky = 1
                                mma(acc[0], wfrag(4, ks), bm0);                                       // (1,1) x a[m][n]
                                mma(acc[1], wfrag(3, ks), bm1); mma(acc[1], wfrag(5, ks), bm0);       // (1,0) x a[m][n+1], (1,2) x a[m][n]
                            } else {            // oy = 2m + 1: ky = 0 reads row m + 1, ky = 2 row m
                                const Frag<T> bn0 = bfrag(row1, 0, ks), bn1 = bfrag(row1, 1, ks);
                                mma(acc[0], wfrag(1, ks), bn0); mma(acc[0], wfrag(7, ks), bm0);       // (0,1), (2,1)
                                mma(acc[1], wfrag(0, ks), bn1); mma(acc[1], wfrag(2, ks), bn0);       // (0,0), (0,2)
                                mma(acc[1], wfrag(6, ks), bm1); mma(acc[1], wfrag(8, ks), bm0);       // (2,0), (2,2)
                            }
                            __builtin_amdgcn_sched_barrier(0);
                        }
                        // epilogue: bias, round to storage, statistics of the rounded values, output order in the wave's LDS tile
                        f32x2 bia[8];
#pragma unroll
                        for (int e = 0; e < 8; ++e) bia[e] = *reinterpret_cast<const f32x2*>(biap + 2 * (e & 1) + 8 * (e >> 1));
#pragma unroll
                        for (int q = 0; q < 2; ++q) {
                            char* cell = otile + (2 * r + q) * OPITCH + 8 * h;
#pragma unroll
                            for (int g = 0; g < 4; ++g) {
                                T4 o4;
#pragma unroll
                                for (int e2 = 0; e2 < 2; ++e2) {
                                    const int e = 2 * g + e2;
                                    const T o0 = (T)(acc[q][2 * e] + bia[e].x), o1 = (T)(acc[q][2 * e + 1] + bia[e].y);
                                    const f32x2 v = f32x2{(float)o0, (float)o1};
                                    s1[e] += v;
                                    s2[e] = f32x2{__builtin_fmaf(v.x, v.x, s2[e].x), __builtin_fmaf(v.y, v.y, s2[e].y)};
                                    o4[2 * e2] = o0; o4[2 * e2 + 1] = o1;
                                }
                                *reinterpret_cast<T4*>(cell + g * 16) = o4;
                            }
                        }
                        // (wave-private tile: LDS executes the wave's accesses in order) 64 output pixels x 64 B = one contiguous 4 KiB run
                        char* dg = reinterpret_cast<char*>(orow_g + (size_t)py_ * (2 * WL) * 32);
#pragma unroll
                        for (int u = 0; u < 8; ++u) {
                            const int id = lane + 64 * u, opx = id >> 3, pc = id & 7;
                            const T4 v = *reinterpret_cast<const T4*>(otile + opx * OPITCH + pc * 8);
                            *reinterpret_cast<T4*>(dg + id * 8) = v;
                        }
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
                const int c = acc_row(2 * e, lane);
                red[wave * 64 + c] = s1[e].x; red[wave * 64 + c + 1] = s1[e].y;
                red[wave * 64 + 32 + c] = s2[e].x; red[wave * 64 + 32 + c + 1] = s2[e].y;
            }
        }
    }
    deep::barrier_lds();
    if (tid < 64 && a.stat) {
        float v = 0.f;
#pragma unroll
        for (int w = 0; w < 8; ++w) v += red[w * 64 + tid];
        unsafeAtomicAdd(&a.stat[stat_rep() * 64 + tid], (double)v);
    }
}
