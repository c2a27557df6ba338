// Workgroup-specialised implicit-GEMM kernels for the DEEP layers of the VAE (encoder.2/3, decoder.0/1; 16-bit storage),
// gfx950.  Same math, operands and epilogues as down2_kernel / up2_kernel (conv_pipe.cuh); what changes is who does what:
//
//   * 512 threads = 4 CONSUMER waves (one per SIMD: MFMA only) + 4 PRODUCER waves (one per SIMD: staging only).  The
//     producers load the raw tensor, apply the per-channel BatchNorm(+LeakyReLU) / BatchNorm-backward map and write the
//     next K chunk's patch into the other half of a double-buffered LDS image WHILE the consumers multiply the current
//     one: VALU and matrix pipe of a SIMD work at the same time instead of one after the other (the one-tile-per-
//     workgroup kernels cost the SUM of their phases: DESIGN.md, phase stamps of round 2);
//   * the weights no longer stream through the vector-memory pipe of every wave: a K step's B operand (already packed
//     [tap][K/8][N][8] by pack_kernel) is copied global -> LDS by LDS-DMA (global_load_lds_dwordx4: no registers, no VALU),
//     one step ahead, once per workgroup, and read back as conflict-free ds_read_b128 fragments by the four consumers;
//   * each consumer owns a 64-pixel x 64-channel (down) / 64-pixel x 32-channel x 4-parity (up) register tile: every
//     fragment read from LDS feeds two MFMAs;
//   * the patch image keeps 80 B per pixel (32 channels + 16 B pad) and pads its rows so that a 32-pixel fragment read
//     touches 16 distinct 16-byte bank groups; the stride-2 kernels keep even and odd columns in separate planes so that a
//     fragment's pixels are consecutive.  Tap and k-step offsets are immediates of the LDS reads.
// One raw s_barrier per K step for all eight waves (waits are explicit: lgkmcnt for LDS traffic, vmcnt for the DMA).
#pragma once
#include "conv_pipe.cuh"

template <typename T> struct DeepConvArgs {
    ConvArgs<T> c;
    int rowp, halfw, imgp, npl;      // LDS patch geometry: pixels per (padded) row, even-column count, pixels per image, pixels in all
    unsigned m_rowp, m_imgp;         // fastdiv magics of rowp / imgp
    int ablate;                      // diagnostic builds (make STAMPS=1) only: bit 0 no weight DMA after step 0, 1 producers store raw, 2 producers idle
};

// make STAMPS=1 (diagnostic build): per-wave cycle counters of the kernels below into ConvArgs::dbg, 16 slots per wave:
//  0 prologue  1 pipeline fill (consumers: interval 0)  2 work (consumers: fragment reads + MFMA; producers: transform + stores)
//  3 waiting at the step barrier  4 epilogue  5 whole kernel  6 waiting at the tile-end rendezvous  7 setup before the BatchNorm
//  finalisation  8 consumers: issuing the weight DMA  9 consumers: waiting for their DMA to land
#ifdef VAE_PHASE_STAMPS
#define DABLATE(da_, bit) (((da_).ablate >> (bit)) & 1)
#define DSTAMP_DECL long long dst_[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}; const long long dst_entry_ = clock64(); long long dst_t_ = dst_entry_;
#define DSTAMP(k) { __builtin_amdgcn_sched_barrier(0); const long long t1_ = clock64(); dst_[k] += t1_ - dst_t_; dst_t_ = t1_; __builtin_amdgcn_sched_barrier(0); }
#define DSTAMP_OUT(a_) { if ((a_).dbg && lane == 0) { dst_[5] = clock64() - dst_entry_; for (int k_ = 0; k_ < 16; ++k_) (a_).dbg[((size_t)blockIdx.x * 8 + wave) * 16 + k_] = dst_[k_]; } }
#else
#define DABLATE(da_, bit) 0
#define DSTAMP_DECL
#define DSTAMP(k)
#define DSTAMP_OUT(a_)
#endif

namespace deep {
// LDS byte offset of 16-byte quarter q of patch pixel j: 64 B of channels + 16 B pad per pixel.  With the padded pitch the
// 16 pixels of a ds_read_b128 lane group land on 16 distinct bank groups when their indices are distinct mod 16 (the launcher
// pads the patch rows for that), and every tap / k-step offset is an immediate of the read.
static constexpr int PPITCH = 80;
__device__ __forceinline__ int px_off(int j, int q) { return j * PPITCH + (q << 4); }
// all LDS traffic of this wave landed, then the workgroup barrier (raw: no vmcnt wait, global loads stay in flight)
__device__ __forceinline__ void barrier_lds() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
// ... and this wave's LDS-DMA copies landed
__device__ __forceinline__ void barrier_dma() { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
// LDS-DMA: 64 lanes x 16 B from per-lane global addresses to 1 KiB of LDS at the wave-uniform address lds_dst.  Issued through
// inline asm on purpose: with the builtin (__builtin_amdgcn_global_load_lds) anywhere in the function hipcc (ROCm 7.2) stops
// counting LDS reads and waits lgkmcnt(0) in front of every MFMA, which exposes the whole LDS latency once per k-step (measured:
// 290 cycles per k-step of 128 MFMA cycles).  The copies are invisible to the compiler's vmcnt bookkeeping: the issuing wave
// waits for them itself (barrier_dma).  M0 (the LDS destination) is saved and restored inside the statement.
__device__ __forceinline__ void dma16(const void* gsrc, char* lds_dst) {
    unsigned keep;
    const unsigned l = (unsigned)(unsigned long long)(__attribute__((address_space(3))) char*)lds_dst;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(l) : "memory");
}
}  // namespace deep

// ---------------------------------------------------------------------------------------------------------------
// Stride-2 "down" product (Conv2d forward, ConvTranspose2d input gradient): 128 low-res pixels x 128 channels per
// workgroup tile, K = 9 taps x Cin walked as (32-channel chunk, tap row, tap column); one barrier interval = one tap row
// of one chunk = 24 MFMAs per consumer.  Tile geometries: 8x16 pixels of one image, or two 8x8 images.
template <typename T, int EPI>
__global__ __launch_bounds__(512) void dn3_kernel(DeepConvArgs<T> da, int n_pairs, int ntiles_n) {
    const ConvArgs<T>& a = da.c;
    constexpr bool TWO_SRC = EPI != EPI_FWD;
    constexpr int CK = 32, NT = 4, MTW = 2, NTW = 2, OROWS = 64;
    constexpr int OPITCH = 32 * NTW * 2 + 16, OCH = 32 * NTW * 2 / 16, OPL = OROWS * OCH / 64;   // out-tile row pitch / chunks per row / chunks per lane
    constexpr int WSLOT = 3 * 4 * 128 * 16;                                                      // one tap row of one chunk: [kx][k/8][n][8]
    constexpr int NE = 8;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), r = lane & 31, h = lane >> 5;
    const int vb = a.xcd ? (int)(blockIdx.x & 7) * a.xcd + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
    const int G = gridDim.x;
    const int th = 1 << a.lth, tw = 1 << a.ltw;
    const int PW = 2 * tw + 1;
    const int Hin = 2 * a.Hs, Win = 2 * a.Ws, Cin = a.Cin, Cout = a.Cout, NCH = Cin / CK;
    const int rowp = da.rowp, halfw = da.halfw, npl = da.npl;
    float* cf = reinterpret_cast<float*>(smem);
    char* patch0 = smem + ((3 * Cin * 4 + 15) & ~15);
    const int PBYTES = npl * deep::PPITCH;
    char* wbuf = patch0 + 2 * PBYTES;
    float* red = reinterpret_cast<float*>(wbuf + 2 * WSLOT);

    const int npm = vb < n_pairs ? (n_pairs - vb + G - 1) / G : 0;   // pairs of this workgroup: vb, vb + G, ...
    if (npm == 0) return;                                            // (uniform; workgroup 0 always has work)
    const int KT = npm * NCH;                                        // K chunks this workgroup walks
    const int n0 = (vb % ntiles_n) * 32 * NT;                        // its N tile (the launcher keeps G a multiple of ntiles_n)

    DSTAMP_DECL
    if (wave >= 4) {
        // =========================== producers ===========================
        if (DABLATE(da, 3)) __builtin_amdgcn_s_setprio(2);
        const int pt = tid - 256, q = pt & 3;
        // the 12 patch pixels (LDS order) this thread stages for every chunk: j = (pt >> 2) + 64 * slot
        int rel[12], flg[12];   // element offset relative to the tile origin; bit 12 unused slot, 13 top halo, 14 left halo, 15.. image
#pragma unroll
        for (int s = 0; s < 12; ++s) {
            const int j = (pt >> 2) + 64 * s;
            const int img = fastdiv(j, da.m_imgp), rem = j - img * da.imgp, py = fastdiv(rem, da.m_rowp), col = rem - py * rowp;
            const int px = col < halfw ? 2 * col : 2 * (col - halfw) + 1;
            const bool used = j < npl && px < PW;
            rel[s] = ((img * Hin + py) * Win + px) * Cin + q * 8;
            flg[s] = (used ? 0 : 1 << 12) | ((py == 0) << 13) | ((px == 0) << 14) | (img << 15);
        }
        auto tile_base = [&](const TileGeo& g, int c0) __attribute__((always_inline)) {
            return ((g.b0 * Hin + 2 * g.y0 - 1) * Win + 2 * g.x0 - 1) * Cin + c0;
        };
        auto item_ok = [&](const TileGeo& g, int f) __attribute__((always_inline)) {
            const int tmask = (1 << 12) | (g.y0 == 0 ? 1 << 13 : 0) | (g.x0 == 0 ? 1 << 14 : 0);
            return ((f & tmask) == 0) & ((f >> 15) < a.B - g.b0);
        };
        Vec16<T> R0[3][4], R1[TWO_SRC ? 3 : 1][TWO_SRC ? 4 : 1];
        auto issue_part = [&](int p, const TileGeo& g, int c0, bool have) __attribute__((always_inline)) {
            const int base = tile_base(g, c0);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int s = p * 4 + u;
                const uint32_t gi = (have & item_ok(g, flg[s])) ? (uint32_t)(base + rel[s]) : 0u;
                R0[p][u] = *reinterpret_cast<const Vec16<T>*>(at_bytes(a.src0, gi * 2u));
                if constexpr (TWO_SRC) R1[p][u] = *reinterpret_cast<const Vec16<T>*>(at_bytes(a.src1, gi * 2u));
            }
        };
        TileGeo cur = decode_pair(a, vb, ntiles_n, 32 * NT);
#pragma unroll
        for (int p = 0; p < 3; ++p) issue_part(p, cur, 0, true);     // chunk 0: in flight during the BatchNorm finalisation below
        DSTAMP(7)

        // (the staging coefficients are derived by the consumer half of the workgroup meanwhile: BatchNorm finalisation below)
        deep::barrier_lds();                                          // coefficients published
        DSTAMP(0)

        f32x2 k0[NE / 2], k1[TWO_SRC ? NE / 2 : 1], k2[NE / 2];
        auto load_coefs = [&](int c0) __attribute__((always_inline)) {
            const int cb = c0 + q * 8;
#pragma unroll
            for (int e = 0; e < NE / 2; ++e) {
                k0[e] = f32x2{cf[cb + 2 * e], cf[cb + 2 * e + 1]}; k2[e] = f32x2{cf[2 * Cin + cb + 2 * e], cf[2 * Cin + cb + 2 * e + 1]};
                if constexpr (TWO_SRC) k1[e] = f32x2{cf[Cin + cb + 2 * e], cf[Cin + cb + 2 * e + 1]};
            }
        };
        auto xform = [&](const Vec16<T>& v0, const Vec16<T>& v1) __attribute__((always_inline)) {
            Vec16<T> o;
#pragma unroll
            for (int e = 0; e < NE / 2; ++e) {
                const f32x2 x0 = {v0.get(2 * e), v0.get(2 * e + 1)};
                f32x2 z;
                if constexpr (TWO_SRC) {
                    const f32x2 x1 = {v1.get(2 * e), v1.get(2 * e + 1)};
                    z = x0 * k0[e] + (x1 * k1[e] + k2[e]);
                } else {
                    z = x0 * k0[e] + k2[e];
                    const f32x2 zs = z * a.slope;
                    z.x = fmaxf(z.x, zs.x); z.y = fmaxf(z.y, zs.y);
                }
                o.set(2 * e, z.x); o.set(2 * e + 1, z.y);
            }
            return o;
        };
        // every slot is stored unconditionally (straight-line code: the compiler then counts the prefetched loads instead of draining
        // them); slots beyond the patch half go to a per-lane dummy cell in the statistics scratch, padding columns are never read
        char* const dummy = reinterpret_cast<char*>(red) + (pt & 63) * 16;
        int c = 0, pj = 0;                                           // chunk within the pair, pair counter
        for (int kk = 0; kk < KT; ++kk) {
            // the chunk after this one
            int cn = c + 1, pjn = pj; TileGeo nxt = cur;
            if (cn == NCH) { cn = 0; pjn = pj + 1; if (pjn < npm) nxt = decode_pair(a, vb + pjn * G, ntiles_n, 32 * NT); }
            const bool nhave = kk + 1 < KT;
            char* pb = patch0 + (kk & 1) * PBYTES + deep::px_off(pt >> 2, q);
            load_coefs(c * CK);
#pragma unroll
            for (int p = 0; p < 3; ++p) {
                if (!DABLATE(da, 2)) {
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int s = p * 4 + u, j = (pt >> 2) + 64 * s;
                    Vec16<T> o = DABLATE(da, 1) ? R0[p][u] : xform(R0[p][u], R1[TWO_SRC ? p : 0][TWO_SRC ? u : 0]);
                    if (!item_ok(cur, flg[s])) o = zero_vec16<T>();
                    *reinterpret_cast<Vec16<T>*>(j < npl ? pb + s * 64 * deep::PPITCH : dummy) = o;
                }
                issue_part(p, nxt, cn * CK, nhave);
                }
                DSTAMP(2)
                // the consumers finish a tile in this interval: one more rendezvous (their epilogue borrows the patch half they just read)
                if (p == 2 && kk >= 1 && c == 0) { deep::barrier_lds(); DSTAMP(6) }
                deep::barrier_lds();
                DSTAMP(3)
            }
            c = cn; pj = pjn; cur = nxt;
        }
        // interval KT: the consumers multiply the last chunk and write the last tile
        deep::barrier_lds(); deep::barrier_lds(); deep::barrier_lds(); deep::barrier_lds();
        DSTAMP(4)
        DSTAMP_OUT(a)
        if constexpr (EPI != EPI_PLAIN) deep::barrier_lds();   // (the consumers' statistics rendezvous below)
        return;
    }

    // =========================== consumers ===========================
    if (!DABLATE(da, 3)) __builtin_amdgcn_s_setprio(1);
    // BatchNorm finalisation of the input layer (workgroup 0 also records it): done by the consumers alone, first thing, while the
    // producers set up their slot tables and get the first chunk's loads in flight
    if (a.fuse.mode != BNF_NONE) {
        for (int i = tid; i < Cin; i += 256) bn_fused_channel(a.fuse, i, blockIdx.x == 0, cf[i], cf[Cin + i], cf[2 * Cin + i]);
    } else {
        for (int i = tid; i < 3 * Cin; i += 256) cf[i] = a.coef[i];
    }
    DSTAMP(7)
    const int wm = wave & 1, wn = wave >> 1, mrow0 = wm * OROWS;
    // weights of K step s (chunk c, tap row ky): 24 pieces of 1 KiB, six per consumer wave
    auto dma_step = [&](int c, int ky, int slot) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const int id = wave * 6 + i, half = id & 1, kg = (id >> 1) & 3, kx = id >> 3;
            int cc = c, kyy = ky;
            if (DABLATE(da, 4)) { cc = (c + vb) % NCH; kyy = (ky + vb / NCH) % 3; }   // timing experiment: workgroups walk the weights out of phase
            const uint32_t off = ((uint32_t)((kyy * 3 + kx) * (Cin >> 3) + cc * 4 + kg) * Cout + n0 + half * 64 + lane) * 16u;
            deep::dma16(at_bytes(a.wp, off), wbuf + slot * WSLOT + id * 1024);
        }
    };
    dma_step(0, 0, 0);
    DSTAMP(10)

    int jb[MTW];
#pragma unroll
    for (int mt = 0; mt < MTW; ++mt) {
        const int R = mrow0 + mt * 32 + r;
        jb[mt] = deep::px_off((R >> (a.lth + a.ltw)) * da.imgp + 2 * ((R >> a.ltw) & (th - 1)) * rowp + (R & (tw - 1)), h);   // byte offset of this lane's pixel, k half h
    }
    float ebv[NTW], esc[NTW], esh[NTW], eis[NTW], exm[NTW];
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt) {
        const int n = n0 + (wn * NTW + nt) * 32 + r;
        ebv[nt] = (EPI == EPI_FWD && a.bias) ? a.bias[n] : 0.f;
        esc[nt] = esh[nt] = eis[nt] = exm[nt] = 0.f;
        if constexpr (EPI == EPI_BWD) {
            esc[nt] = a.ocoef[LC_SC * Cout + n]; esh[nt] = a.ocoef[LC_SH * Cout + n];
            eis[nt] = a.ocoef[LC_INVSTD * Cout + n]; exm[nt] = a.ocoef[LC_XM * Cout + n];
        }
    }
    f32x16 acc[MTW][NTW];
    f32x2 s1[NTW], s2[NTW];
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt) {
        s1[nt] = f32x2{0.f, 0.f}; s2[nt] = f32x2{0.f, 0.f};
#pragma unroll
        for (int mt = 0; mt < MTW; ++mt)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[mt][nt][i] = ebv[nt];
    }
    DSTAMP(11)
    // out-tile chunk u of this lane (wave-local chunk id = lane + 64u)
    int orel[OPL], opk[OPL];
#pragma unroll
    for (int u = 0; u < OPL; ++u) {
        const int id = lane + 64 * u, row = id / OCH, qq = id - row * OCH, RR = mrow0 + row;
        const int img = RR >> (a.lth + a.ltw), ty = (RR >> a.ltw) & (th - 1), tx = RR & (tw - 1);
        orel[u] = ((img * a.Hs + ty) * a.Ws + tx) * Cout + wn * NTW * 32 + qq * 8;
        opk[u] = (row * OPITCH + qq * 16) | (img << 20);
    }
    decltype(Vec16<T>::v) prey[EPI == EPI_BWD ? OPL : 1];
    DSTAMP(12)

    deep::barrier_lds();

    TileGeo cur = decode_pair(a, vb, ntiles_n, 32 * NT);
    int c = 0, pj = 0;
    // interval (kk, p): the producers stage chunk kk; the consumers multiply chunk kk - 1, tap row p
    DSTAMP(0)
    deep::barrier_lds(); deep::barrier_lds(); deep::barrier_dma();   // kk = 0: chunk 0 is being staged; the weights of step 0 have landed
    DSTAMP(1)
    for (int kk = 1; kk <= KT; ++kk) {
        const char* pb = patch0 + ((kk - 1) & 1) * PBYTES;
        const bool last_chunk = c == NCH - 1;
#pragma unroll
        for (int p = 0; p < 3; ++p) {
            const int s = 3 * (kk - 1) + p;
            // weights of the next step, into the slot the previous step has just released
            if (!DABLATE(da, 0)) {
                if (p < 2) dma_step(c, p + 1, (s + 1) & 1);
                else if (kk < KT) dma_step(c + 1 == NCH ? 0 : c + 1, 0, (s + 1) & 1);
            }
            if constexpr (EPI == EPI_BWD) {
                if (last_chunk && p == 1) {   // y_out rows of this tile for the epilogue, one interval ahead
                    const int base = ((cur.b0 * a.Hs + cur.y0) * a.Ws + cur.x0) * Cout + n0;
#pragma unroll
                    for (int u = 0; u < OPL; ++u) {
                        const int gi = (cur.b0 + (opk[u] >> 20)) < a.B ? base + orel[u] : 0;
                        prey[u] = *reinterpret_cast<const decltype(Vec16<T>::v)*>(at_bytes(a.yout, (uint32_t)gi * 2u));
                    }
                }
            }
            const char* wb = wbuf + (s & 1) * WSLOT;
            DSTAMP(8)
            {
                Frag<T> af[2][MTW], bf[2][NTW];
                auto load_frags = [&](int st, int buf) __attribute__((always_inline)) {
                    const int kx = st >> 1, ks = st & 1;
                    const int koff = (p * rowp + (kx == 0 ? 0 : (kx == 1 ? halfw : 1))) * deep::PPITCH + ks * 32;
#pragma unroll
                    for (int mt = 0; mt < MTW; ++mt) af[buf][mt] = load_frag(reinterpret_cast<const T*>(pb + jb[mt] + koff));
#pragma unroll
                    for (int nt = 0; nt < NTW; ++nt)
                        bf[buf][nt] = load_frag(reinterpret_cast<const T*>(wb + (((kx * 4 + ks * 2 + h) * 128 + (wn * NTW + nt) * 32 + r) << 4)));
                };
                // the fragments of k-step st + 1 are requested BEFORE the MFMAs of k-step st (scheduling fences: left to itself the
                // compiler sinks every read to just in front of its use and exposes the LDS latency twice per k-step)
                load_frags(0, 0);
#pragma unroll
                for (int st = 0; st < 6; ++st) {
                    if (st + 1 < 6) load_frags(st + 1, (st + 1) & 1);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
                        for (int mt = 0; mt < MTW; ++mt) mma(acc[mt][nt], af[st & 1][mt], bf[st & 1][nt]);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            DSTAMP(2)
            if (p == 2 && last_chunk) {
                // ---- epilogue through a wave-private tile borrowed from the patch half just consumed
                deep::barrier_lds();                                  // every consumer is done reading it
                DSTAMP(6)
                char* mytile = const_cast<char*>(pb) + wave * OROWS * OPITCH;
                const int obase = ((cur.b0 * a.Hs + cur.y0) * a.Ws + cur.x0) * Cout + n0;
                if constexpr (EPI == EPI_BWD) {
#pragma unroll
                    for (int u = 0; u < OPL; ++u) *reinterpret_cast<decltype(Vec16<T>::v)*>(mytile + (opk[u] & 0xfffff)) = prey[u];
                }
                auto epi_body = [&](auto checked) __attribute__((always_inline)) {
#pragma unroll
                    for (int nt = 0; nt < NTW; ++nt) {
#pragma unroll
                        for (int mt = 0; mt < MTW; ++mt) {
#pragma unroll
                            for (int i = 0; i < 16; i += 2) {
                                const int row0 = mt * 32 + acc_row(i, lane), row1 = mt * 32 + acc_row(i + 1, lane);
                                T* c0 = reinterpret_cast<T*>(mytile + row0 * OPITCH) + nt * 32 + r;
                                T* c1 = reinterpret_cast<T*>(mytile + row1 * OPITCH) + nt * 32 + r;
                                constexpr bool CK_ = decltype(checked)::value;
                                const bool ok0 = !CK_ || (cur.b0 + ((mrow0 + row0) >> (a.lth + a.ltw))) < a.B;
                                const bool ok1 = !CK_ || (cur.b0 + ((mrow0 + row1) >> (a.lth + a.ltw))) < a.B;
                                epi_pair<T, EPI, CK_>(acc[mt][nt][i], acc[mt][nt][i + 1], c0, c1, ok0, ok1, esc[nt], esh[nt], a.oslope, s1[nt], s2[nt]);
                                acc[mt][nt][i] = ebv[nt]; acc[mt][nt][i + 1] = ebv[nt];
                            }
                        }
                    }
                };
                if (EPI != EPI_FWD || cur.b0 + (1 << a.lTB) <= a.B) epi_body(std::false_type{}); else epi_body(std::true_type{});
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
                for (int u = 0; u < OPL; ++u) {
                    const Vec16<T> v = *reinterpret_cast<const Vec16<T>*>(mytile + (opk[u] & 0xfffff));
                    if ((cur.b0 + (opk[u] >> 20)) < a.B) *reinterpret_cast<Vec16<T>*>(at_bytes(a.out, (uint32_t)(obase + orel[u]) * 2u)) = v;
                }
                DSTAMP(4)
            }
#ifdef VAE_PHASE_STAMPS
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            DSTAMP(9)
#endif
            deep::barrier_dma();
            DSTAMP(3)
        }
        if (++c == NCH) { c = 0; ++pj; if (pj < npm) cur = decode_pair(a, vb + pj * G, ntiles_n, 32 * NT); }
    }
    DSTAMP_OUT(a)

    if constexpr (EPI != EPI_PLAIN) {
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt) {
            float v1 = s1[nt].x + s1[nt].y, v2 = s2[nt].x + s2[nt].y;
            if constexpr (EPI == EPI_BWD) v2 = eis[nt] * v2 + exm[nt] * v1;   // sum dz*xhat from sum dz*y and sum dz
            v1 += __shfl_xor(v1, 32, 64); v2 += __shfl_xor(v2, 32, 64);
            if (h == 0) { red[((wave * NTW + nt) * 32 + r) * 2] = v1; red[((wave * NTW + nt) * 32 + r) * 2 + 1] = v2; }
        }
        deep::barrier_lds();
        if (tid < NT * 32) {   // channel tid of the tile lives in the two waves of column wn = tid / 64
            const int cw = tid / (32 * NTW), cl = tid - cw * 32 * NTW;
            float v1 = 0.f, v2 = 0.f;
#pragma unroll
            for (int m = 0; m < 2; ++m) { v1 += red[(((m + 2 * cw) * NTW) * 32 + cl) * 2]; v2 += red[(((m + 2 * cw) * NTW) * 32 + cl) * 2 + 1]; }
            double* st_ = a.stat + stat_rep() * 2 * Cout;
            unsafeAtomicAdd(&st_[n0 + tid], (double)v1);
            unsafeAtomicAdd(&st_[Cout + n0 + tid], (double)v2);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Stride-2 "up" product (ConvTranspose2d forward, Conv2d input gradient): 128 low-res pixels x 64 output channels x the
// four output parities per workgroup tile; K = Cin walked in 32-channel chunks with all nine taps per chunk; one barrier
// interval = one chunk = 36 MFMAs per consumer (64 low-res pixels x 32 channels x 4 parities: 8 accumulator tiles).
// Parity class of a tap and the input pixel offset it reads are those of up2_kernel (conv_pipe.cuh).
template <typename T, int EPI>
__global__ __launch_bounds__(512) void up3_kernel(DeepConvArgs<T> da, int n_pairs, int ntiles_n) {
    const ConvArgs<T>& a = da.c;
    constexpr bool TWO_SRC = EPI != EPI_FWD;
    constexpr int CK = 32, MTW = 2, NS = 7;                            // NS: patch pixels a producer thread stages per chunk
    constexpr int OPITCH = 64 + 16, OCH = 4, OPL = 4;                  // epilogue round: 64 out pixels x 32 channels per wave
    constexpr int WSLOT = 9 * 4 * 64 * 16;                             // one chunk of weights: [tap][k/8][n][8]
    constexpr int NE = 8, NTAP = 9;
    constexpr int tap_cls[NTAP] = {0, 1, 1, 2, 2, 3, 3, 3, 3};
    constexpr int tap_off[NTAP] = {0, 0, 1, 0, 2, 0, 1, 2, 3};         // di*2+dj
    constexpr unsigned long long TAP_T = 0x026817354ULL;               // tap_t[k] = {4,5,3,7,1,8,6,2,0}, one nibble each
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), r = lane & 31, h = lane >> 5;
    const int vb = a.xcd ? (int)(blockIdx.x & 7) * a.xcd + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
    const int G = gridDim.x;
    const int th = 1 << a.lth, tw = 1 << a.ltw;
    const int PW = tw + 1;
    const int Hs = a.Hs, Ws = a.Ws, Cin = a.Cin, Cout = a.Cout, NCH = Cin / CK;
    const int rowp = da.rowp, npl = da.npl;
    float* cf = reinterpret_cast<float*>(smem);
    char* patch0 = smem + ((3 * Cin * 4 + 15) & ~15);
    const int PBYTES = npl * deep::PPITCH;
    char* wbuf = patch0 + 2 * PBYTES;
    float* red = reinterpret_cast<float*>(wbuf + 2 * WSLOT);

    const int npm = vb < n_pairs ? (n_pairs - vb + G - 1) / G : 0;
    if (npm == 0) return;
    const int KT = npm * NCH;
    const int n0 = (vb % ntiles_n) * 64;

    if (wave >= 4) {
        // =========================== producers ===========================
        const int pt = tid - 256, q = pt & 3;
        int rel[NS], flg[NS];   // bit 12 unused slot, 13 bottom halo, 14 right halo, 15.. image
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const int j = (pt >> 2) + 64 * s;
            const int img = fastdiv(j, da.m_imgp), rem = j - img * da.imgp, py = fastdiv(rem, da.m_rowp), px = rem - py * rowp;
            const bool used = j < npl && px < PW;
            rel[s] = ((img * Hs + py) * Ws + px) * Cin + q * 8;
            flg[s] = (used ? 0 : 1 << 12) | ((py == th) << 13) | ((px == tw) << 14) | (img << 15);
        }
        auto tile_base = [&](const TileGeo& g, int c0) __attribute__((always_inline)) {
            return ((g.b0 * Hs + g.y0) * Ws + g.x0) * Cin + c0;
        };
        auto item_ok = [&](const TileGeo& g, int f) __attribute__((always_inline)) {
            const int tmask = (1 << 12) | (g.y0 + th >= Hs ? 1 << 13 : 0) | (g.x0 + tw >= Ws ? 1 << 14 : 0);
            return ((f & tmask) == 0) & ((f >> 15) < a.B - g.b0);
        };
        Vec16<T> R0[2][NS], R1[TWO_SRC ? 2 : 1][TWO_SRC ? NS : 1];    // two chunks in flight
        auto issue_chunk = [&](int buf, const TileGeo& g, int c0, bool have) __attribute__((always_inline)) {
            const int base = tile_base(g, c0);
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                const uint32_t gi = (have & item_ok(g, flg[s])) ? (uint32_t)(base + rel[s]) : 0u;
                R0[buf][s] = *reinterpret_cast<const Vec16<T>*>(at_bytes(a.src0, gi * 2u));
                if constexpr (TWO_SRC) R1[buf][s] = *reinterpret_cast<const Vec16<T>*>(at_bytes(a.src1, gi * 2u));
            }
        };
        // chunk sequence of this workgroup: chunk index k -> (pair k / NCH, channel chunk k % NCH)
        auto chunk_geo = [&](int k, TileGeo& g, int& c) __attribute__((always_inline)) {
            const int pj = k / NCH; c = k - pj * NCH;
            g = decode_pair(a, vb + pj * G, ntiles_n, 64);
        };
        {
            TileGeo g0, g1; int c0_, c1_;
            chunk_geo(0, g0, c0_); issue_chunk(0, g0, 0, true);
            chunk_geo(KT > 1 ? 1 : 0, g1, c1_); issue_chunk(1, g1, c1_ * CK, KT > 1);
        }
        // (the staging coefficients are derived by the consumer half of the workgroup meanwhile: BatchNorm finalisation below)
        deep::barrier_lds();

        f32x2 k0[NE / 2], k1[TWO_SRC ? NE / 2 : 1], k2[NE / 2];
        auto load_coefs = [&](int c0) __attribute__((always_inline)) {
            const int cb = c0 + q * 8;
#pragma unroll
            for (int e = 0; e < NE / 2; ++e) {
                k0[e] = f32x2{cf[cb + 2 * e], cf[cb + 2 * e + 1]}; k2[e] = f32x2{cf[2 * Cin + cb + 2 * e], cf[2 * Cin + cb + 2 * e + 1]};
                if constexpr (TWO_SRC) k1[e] = f32x2{cf[Cin + cb + 2 * e], cf[Cin + cb + 2 * e + 1]};
            }
        };
        auto xform = [&](const Vec16<T>& v0, const Vec16<T>& v1) __attribute__((always_inline)) {
            Vec16<T> o;
#pragma unroll
            for (int e = 0; e < NE / 2; ++e) {
                const f32x2 x0 = {v0.get(2 * e), v0.get(2 * e + 1)};
                f32x2 z;
                if constexpr (TWO_SRC) {
                    const f32x2 x1 = {v1.get(2 * e), v1.get(2 * e + 1)};
                    z = x0 * k0[e] + (x1 * k1[e] + k2[e]);
                } else {
                    z = x0 * k0[e] + k2[e];
                    const f32x2 zs = z * a.slope;
                    z.x = fmaxf(z.x, zs.x); z.y = fmaxf(z.y, zs.y);
                }
                o.set(2 * e, z.x); o.set(2 * e + 1, z.y);
            }
            return o;
        };
        char* const dummy = reinterpret_cast<char*>(red) + (pt & 63) * 16;
        // interval kk: stage chunk kk (registers of buffer kk & 1) into patch half kk & 1, then request chunk kk + 2
        auto stage = [&](int kk, auto bufc) __attribute__((always_inline)) {
            constexpr int BUF = decltype(bufc)::value;
            if (kk < KT) {
                TileGeo g, gn; int c, cn;
                chunk_geo(kk, g, c);
                const bool nhave = kk + 2 < KT;
                chunk_geo(nhave ? kk + 2 : kk, gn, cn);
                load_coefs(c * CK);
                char* pb = patch0 + BUF * PBYTES + deep::px_off(pt >> 2, q);
#pragma unroll
                for (int s = 0; s < NS; ++s) {   // (unconditional stores: see dn3_kernel)
                    const int j = (pt >> 2) + 64 * s;
                    Vec16<T> o = xform(R0[BUF][s], R1[TWO_SRC ? BUF : 0][TWO_SRC ? s : 0]);
                    if (!item_ok(g, flg[s])) o = zero_vec16<T>();
                    *reinterpret_cast<Vec16<T>*>(j < npl ? pb + s * 64 * deep::PPITCH : dummy) = o;
                }
                issue_chunk(BUF, gn, cn * CK, nhave);
                if (kk >= 1 && c == 0) deep::barrier_lds();   // the consumers finish a tile in this interval (their epilogue borrows the weight slot they just read)
            } else if (kk >= 1) deep::barrier_lds();           // kk == KT: the last tile's epilogue
            deep::barrier_lds();
        };
        for (int kk = 0; kk <= KT; kk += 2) {
            stage(kk, std::integral_constant<int, 0>{});
            if (kk + 1 <= KT) stage(kk + 1, std::integral_constant<int, 1>{});
        }
        if constexpr (EPI != EPI_PLAIN) deep::barrier_lds();
        return;
    }

    // =========================== consumers ===========================
    __builtin_amdgcn_s_setprio(1);
    if (a.fuse.mode != BNF_NONE) {
        for (int i = tid; i < Cin; i += 256) bn_fused_channel(a.fuse, i, blockIdx.x == 0, cf[i], cf[Cin + i], cf[2 * Cin + i]);
    } else {
        for (int i = tid; i < 3 * Cin; i += 256) cf[i] = a.coef[i];
    }
    const int wm = wave & 1, wn = wave >> 1, mrow0 = wm * 64;
    auto dma_chunk = [&](int c, int slot) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 9; ++i) {
            const int id = wave * 9 + i, kg = id & 3, k9 = id >> 2;
            const int tap = (int)((TAP_T >> (4 * k9)) & 15);
            const uint32_t off = ((uint32_t)(tap * (Cin >> 3) + c * 4 + kg) * Cout + n0 + lane) * 16u;
            deep::dma16(at_bytes(a.wp, off), wbuf + slot * WSLOT + id * 1024);
        }
    };
    dma_chunk(0, 0);

    int jb[MTW];
#pragma unroll
    for (int mt = 0; mt < MTW; ++mt) {
        const int R = mrow0 + mt * 32 + r;
        jb[mt] = deep::px_off((R >> (a.lth + a.ltw)) * da.imgp + ((R >> a.ltw) & (th - 1)) * rowp + (R & (tw - 1)), h);
    }
    const int n = n0 + wn * 32 + r;
    const float ebv = (EPI == EPI_FWD && a.bias) ? a.bias[n] : 0.f;
    float esc = 0.f, esh = 0.f, eis = 0.f, exm = 0.f;
    if constexpr (EPI == EPI_BWD) {
        esc = a.ocoef[LC_SC * Cout + n]; esh = a.ocoef[LC_SH * Cout + n];
        eis = a.ocoef[LC_INVSTD * Cout + n]; exm = a.ocoef[LC_XM * Cout + n];
    }
    f32x16 acc[4][MTW];
    f32x2 s1 = {0.f, 0.f}, s2 = {0.f, 0.f};
#pragma unroll
    for (int cl = 0; cl < 4; ++cl)
#pragma unroll
        for (int mt = 0; mt < MTW; ++mt)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[cl][mt][i] = ebv;
    // epilogue round (mt, py): wave-local out pixel o = 2*row + px (row = low-res pixel 0..31 of the sub-tile) -> LDS row o.
    // Chunk u of this lane: id = lane + 64u, o = id / 4, quarter id % 4; element offset relative to the tile origin for mt = 0, py = 0
    int orel[OPL], opk[OPL];
#pragma unroll
    for (int u = 0; u < OPL; ++u) {
        const int id = lane + 64 * u, o = id / OCH, qq = id - o * OCH, row = o >> 1, px = o & 1, RR = mrow0 + row;
        const int img = RR >> (a.lth + a.ltw), ty = (RR >> a.ltw) & (th - 1), tx = RR & (tw - 1);
        orel[u] = ((img * 2 * Hs + 2 * ty) * 2 * Ws + 2 * tx + px) * Cout + wn * 32 + qq * 8;
        opk[u] = (o * OPITCH + qq * 16) | (img << 20);
    }
    const int mt_step = (32 >> a.ltw) * 2 * 2 * Ws * Cout;             // second 32-pixel sub-tile: 32 / tw low-res rows further down
    const int py_step = 2 * Ws * Cout;
    decltype(Vec16<T>::v) prey[EPI == EPI_BWD ? OPL : 1];

    deep::barrier_lds();

    TileGeo cur = decode_pair(a, vb, ntiles_n, 64);
    int c = 0, pj = 0;
    deep::barrier_dma();                                               // interval 0: the producers stage chunk 0; weights of chunk 0 landed
    for (int kk = 1; kk <= KT; ++kk) {
        const char* pb = patch0 + ((kk - 1) & 1) * PBYTES;
        const char* wb = wbuf + ((kk - 1) & 1) * WSLOT;
        const bool last_chunk = c == NCH - 1;
        if (kk < KT) dma_chunk(c + 1 == NCH ? 0 : c + 1, kk & 1);
        const int obase = ((cur.b0 * 2 * Hs + 2 * cur.y0) * 2 * Ws + 2 * cur.x0) * Cout + n0;
        auto issue_y = [&](int round) __attribute__((always_inline)) {
            const int rb = obase + (round >> 1) * mt_step + (round & 1) * py_step;
#pragma unroll
            for (int u = 0; u < OPL; ++u) {
                const int gi = (cur.b0 + (opk[u] >> 20)) < a.B ? rb + orel[u] : 0;
                prey[u] = *reinterpret_cast<const decltype(Vec16<T>::v)*>(at_bytes(a.yout, (uint32_t)gi * 2u));
            }
        };
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            Frag<T> af[MTW][4];
#pragma unroll
            for (int mt = 0; mt < MTW; ++mt)
#pragma unroll
                for (int o = 0; o < 4; ++o) {
                    af[mt][o] = load_frag(reinterpret_cast<const T*>(pb + jb[mt] + ((o >> 1) * rowp + (o & 1)) * deep::PPITCH + ks * 32));
                }
            constexpr int DEPTH = 4;
            Frag<T> bq[DEPTH];
            auto load_b = [&](int k9, int slot) __attribute__((always_inline)) {
                bq[slot] = load_frag(reinterpret_cast<const T*>(wb + (((k9 * 4 + ks * 2 + h) * 64 + wn * 32 + r) << 4)));
            };
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) load_b(d, d);
#pragma unroll
            for (int k9 = 0; k9 < NTAP; ++k9) {
#pragma unroll
                for (int mt = 0; mt < MTW; ++mt) mma(acc[tap_cls[k9]][mt], af[mt][tap_off[k9]], bq[k9 % DEPTH]);
                if (k9 + DEPTH < NTAP) load_b(k9 + DEPTH, k9 % DEPTH);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (last_chunk) {
            // ---- epilogue: four rounds (sub-tile mt, output row parity py) through a wave-private tile borrowed from the weight slot just read
            if constexpr (EPI == EPI_BWD) issue_y(0);
            deep::barrier_lds();                                       // every consumer is done reading the slot
            char* mytile = const_cast<char*>(wb) + wave * 64 * OPITCH;
#pragma unroll
            for (int round = 0; round < 4; ++round) {
                const int mt = round >> 1, py = round & 1;
                if constexpr (EPI == EPI_BWD) {
#pragma unroll
                    for (int u = 0; u < OPL; ++u) *reinterpret_cast<decltype(Vec16<T>::v)*>(mytile + (opk[u] & 0xfffff)) = prey[u];
                    if (round < 3) issue_y(round + 1);
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                }
                auto epi_body = [&](auto checked) __attribute__((always_inline)) {
#pragma unroll
                    for (int px = 0; px < 2; ++px) {
#pragma unroll
                        for (int i = 0; i < 16; i += 2) {
                            const int row0 = acc_row(i, lane), row1 = acc_row(i + 1, lane);
                            T* c0 = reinterpret_cast<T*>(mytile + (2 * row0 + px) * OPITCH) + r;
                            T* c1 = reinterpret_cast<T*>(mytile + (2 * row1 + px) * OPITCH) + r;
                            constexpr bool CK_ = decltype(checked)::value;
                            const bool ok0 = !CK_ || (cur.b0 + ((mrow0 + mt * 32 + row0) >> (a.lth + a.ltw))) < a.B;
                            const bool ok1 = !CK_ || (cur.b0 + ((mrow0 + mt * 32 + row1) >> (a.lth + a.ltw))) < a.B;
                            epi_pair<T, EPI, CK_>(acc[py * 2 + px][mt][i], acc[py * 2 + px][mt][i + 1], c0, c1, ok0, ok1, esc, esh, a.oslope, s1, s2);
                            acc[py * 2 + px][mt][i] = ebv; acc[py * 2 + px][mt][i + 1] = ebv;
                        }
                    }
                };
                if (EPI != EPI_FWD || cur.b0 + (1 << a.lTB) <= a.B) epi_body(std::false_type{}); else epi_body(std::true_type{});
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                const int rb = obase + mt * mt_step + py * py_step;
#pragma unroll
                for (int u = 0; u < OPL; ++u) {
                    const Vec16<T> v = *reinterpret_cast<const Vec16<T>*>(mytile + (opk[u] & 0xfffff));
                    if ((cur.b0 + (opk[u] >> 20)) < a.B) *reinterpret_cast<Vec16<T>*>(at_bytes(a.out, (uint32_t)(rb + orel[u]) * 2u)) = v;
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // reads done before the next round overwrites the tile
            }
        }
        deep::barrier_dma();
        if (++c == NCH) { c = 0; ++pj; if (pj < npm) cur = decode_pair(a, vb + pj * G, ntiles_n, 64); }
    }

    if constexpr (EPI != EPI_PLAIN) {
        float v1 = s1.x + s1.y, v2 = s2.x + s2.y;
        if constexpr (EPI == EPI_BWD) v2 = eis * v2 + exm * v1;
        v1 += __shfl_xor(v1, 32, 64); v2 += __shfl_xor(v2, 32, 64);
        if (h == 0) { red[(wave * 32 + r) * 2] = v1; red[(wave * 32 + r) * 2 + 1] = v2; }
        deep::barrier_lds();
        if (tid < 64) {   // channel tid of the tile lives in the two waves of column wn = tid / 32
            const int cw = tid >> 5, cl = tid & 31;
            float t1 = 0.f, t2 = 0.f;
#pragma unroll
            for (int m = 0; m < 2; ++m) { t1 += red[((m + 2 * cw) * 32 + cl) * 2]; t2 += red[((m + 2 * cw) * 32 + cl) * 2 + 1]; }
            double* st_ = a.stat + stat_rep() * 2 * Cout;
            unsafeAtomicAdd(&st_[n0 + tid], (double)t1);
            unsafeAtomicAdd(&st_[Cout + n0 + tid], (double)t2);
        }
    }
}
