// Weight gradient of the deep stride-2 layers (encoder.2/3, decoder.0/1; 16-bit storage, 128 low-res-side channels x 32 high-res-
// side channels x 9 taps per workgroup), gfx950.  Same operands, staging transforms, tile walk and MFMA order as
// wgrad_kernel<T, 4, 1, CONVT, true, 8> (conv_mfma.cuh) - results are bit-identical - but the two halves of that kernel's K loop
// no longer alternate in the same waves:
//
//   * 1024 threads = 8 PRODUCER waves + 8 CONSUMER waves, two of each per SIMD, each group with its own loop (and its own
//     register allocation under the 128-VGPR ceiling of a 16-wave workgroup: the staging half needs the prefetch registers and
//     the per-channel coefficients, the MFMA half the accumulators of 5 taps);
//   * producers: load K tile i+1's operand chunks into registers, apply the BatchNorm(+LeakyReLU) / BatchNorm-backward map to
//     tile i's and store them into LDS buffer i & 1;
//   * consumers: transposed k-major reads + MFMAs of tile i-1 from buffer (i-1) & 1.
//   One raw s_barrier per K tile.  In the 8-wave kernel a K tile cost staging + MFMA (~7.8k cycles of which ~1.2k MFMA at two
//   waves per SIMD, 51 us per layer at 128 workgroups); here it costs the longer of the two with four waves per SIMD hiding each
//   other's latencies.
#pragma once
#include "conv_mfma.cuh"
#include "conv_deep.cuh"

namespace wsp {
static constexpr int WA = 4, NP = 512, SIT = WG_KP * 4 * WA / NP, TS = 2, NTW = 5, MAXG = 3;
template <typename T> struct Geo {
    static constexpr int E16 = 16 / sizeof(T), SROW = 32 * WA * sizeof(T), GROW = 32 * sizeof(T);
    static constexpr int SPITCH = SROW + WG_SPAD, GPITCH = GROW + WG_GPAD, SCH = SROW / 16, GCH = GROW / 16;
};
}
// LDS bytes: coefficient rows, two (low-res tile, high-res patch) buffers, the patch staging table (padded to MAXG * NP entries)
template <typename T> static inline size_t wgrad_split_lds(int npix) {
    typedef wsp::Geo<T> G;
    return (size_t)(3 * 32 * wsp::WA + 3 * 32) * 4 + 2 * ((size_t)WG_KP * G::SPITCH + (size_t)npix * G::GPITCH) +
           std::max<size_t>((size_t)npix * G::GCH, (size_t)wsp::MAXG * wsp::NP) * 8;
}

template <typename T, bool CONVT>
__global__ __launch_bounds__(1024) void wgrad_split_kernel(WgradArgs<T> a) {
    using namespace wsp;
    typedef Geo<T> G;
    constexpr int E16 = G::E16, SPITCH = G::SPITCH, GPITCH = G::GPITCH, SCH = G::SCH, GCH = G::GCH;
    constexpr bool S_TWO = !CONVT, G_TWO = CONVT;
    constexpr int NE = Vec16<T>::N;
    static_assert(sizeof(T) == 2, "16-bit storage only");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), r = lane & 31;
    const int th = 1 << a.lth, tw = 1 << a.ltw, TB = 1 << a.lTB;
    const int PH = 2 * th + 1, PW = 2 * tw + 1, PP = PH * PW, npix = TB * PP;
    const int Hs = a.Hs, Ws = a.Ws, Hg = 2 * a.Hs, Wg = 2 * a.Ws, CA = a.CA, CB = a.CB;
    const int a0 = blockIdx.y * 32 * WA, bc0 = blockIdx.z * 32;

    float* cfs = reinterpret_cast<float*>(smem);             // [3][128]
    float* cfg = cfs + 3 * 32 * WA;                          // [3][32]
    char* stile0 = reinterpret_cast<char*>(cfg + 3 * 32);    // [2][64][SPITCH]
    char* gtile0 = stile0 + 2 * WG_KP * SPITCH;              // [2][npix][GPITCH]
    const int SBUF = WG_KP * SPITCH, GBUF = npix * GPITCH;
    // tile-independent staging table of the high-res patch: {relative element offset, LDS offset/16 | top<<13 | left<<14 | img<<15}
    int2* gtab = reinterpret_cast<int2*>(gtile0 + 2 * GBUF);
    for (int it = tid; it < max(npix * GCH, MAXG * NP); it += 1024) {
        const int pix = it / GCH, qq = it - pix * GCH;
        const int img = fastdiv(pix, a.m_pp), rem = pix - img * PP, py = fastdiv(rem, a.m_pw), px = rem - py * PW;
        gtab[it] = it < npix * GCH ? make_int2(((img * Hg + py) * Wg + px) * CB + qq * E16,
                                               ((pix * GPITCH + qq * 16) >> 4) | ((py == 0) << 13) | ((px == 0) << 14) | (img << 15))
                                   : make_int2(0, 0xffff << 15);
    }
    // staging coefficients (tile-local rows); the gradient operand's may be derived here from the batch statistics
    if (a.fuse.mode == BNF_BWD && a.s_two) {
        for (int i = tid; i < 32 * WA; i += 1024) bn_fused_channel(a.fuse, a0 + i, false, cfs[i], cfs[32 * WA + i], cfs[2 * 32 * WA + i]);
    } else {
        for (int i = tid; i < 3 * 32 * WA; i += 1024) cfs[i] = a.scoef[(i / (32 * WA)) * CA + a0 + (i % (32 * WA))];
    }
    if (a.fuse.mode == BNF_BWD && a.g_two) {
        for (int i = tid; i < 32; i += 1024) bn_fused_channel(a.fuse, bc0 + i, false, cfg[i], cfg[32 + i], cfg[64 + i]);
    } else {
        for (int i = tid; i < 3 * 32; i += 1024) cfg[i] = a.gcoef[(i / 32) * CB + bc0 + (i % 32)];
    }
    const int t_begin = blockIdx.x * a.tiles_per_split;
    const int t_end = min(a.n_tiles, t_begin + a.tiles_per_split), nt = max(0, t_end - t_begin);
    auto tile_origin = [&](int tile_, int& b0, int& y0, int& x0) {
        const int tile = a.rev ? a.n_tiles - 1 - tile_ : tile_;   // reversed walk (see ConvArgs::rev)
        const int bt = fastdiv(tile, a.m_txy), trem = tile - bt * a.tiles_x * a.tiles_y, ty = fastdiv(trem, a.m_tx), tx = trem - ty * a.tiles_x;
        b0 = bt << a.lTB; y0 = ty << a.lth; x0 = tx << a.ltw;
    };
#ifdef VAE_PHASE_STAMPS
    long long dst_[8] = {0, 0, 0, 0, 0, 0, 0, 0}; const long long dst_entry_ = clock64(); long long dst_t_ = dst_entry_;
#define WSTAMP(k) { __builtin_amdgcn_sched_barrier(0); const long long t1_ = clock64(); dst_[k] += t1_ - dst_t_; dst_t_ = t1_; __builtin_amdgcn_sched_barrier(0); }
#else
#define WSTAMP(k)
#endif
    deep::barrier_lds();                   // coefficient rows / table published
    WSTAMP(0)

    if (wave >= 8) {
        // =========================== producers ===========================
        const int pt = tid - 512;
        // A thread always stages the same 16-byte channel quarter of both operands (512 is a multiple of the chunks per pixel), so
        // its coefficients live in registers; offsets are 32-bit bytes.
        Vec16<T> ps0[SIT], ps1[S_TWO ? SIT : 1], pg0[MAXG], pg1[G_TWO ? MAXG : 1];
        int srel[SIT], sloff[SIT];     // low-res operand: tile-independent element offset / LDS offset | image << 20
        int gmeta[MAXG];               // high-res operand: table word of the chunk (LDS offset, halo flags, image)
        f32x2 ks0[NE / 2], ks1[S_TWO ? NE / 2 : 1], ks2[NE / 2];
        f32x2 kg0[NE / 2], kg1[G_TWO ? NE / 2 : 1], kg2[NE / 2];
#pragma unroll
        for (int u = 0; u < SIT; ++u) {
            const int it = pt + u * NP, k = it / SCH, qq = it - k * SCH;
            const int img = k >> (a.lth + a.ltw), y = (k >> a.ltw) & (th - 1), x = k & (tw - 1);
            srel[u] = ((img * Hs + y) * Ws + x) * CA + a0 + qq * E16;
            sloff[u] = (k * SPITCH + qq * 16) | (img << 20);
        }
        const int sq = (pt % SCH) * E16, gq = (pt % GCH) * E16;
#pragma unroll
        for (int e = 0; e < NE / 2; ++e) {
            ks0[e] = f32x2{cfs[sq + 2 * e], cfs[sq + 2 * e + 1]}; ks2[e] = f32x2{cfs[2 * 32 * WA + sq + 2 * e], cfs[2 * 32 * WA + sq + 2 * e + 1]};
            if constexpr (S_TWO) ks1[e] = f32x2{cfs[32 * WA + sq + 2 * e], cfs[32 * WA + sq + 2 * e + 1]};
            kg0[e] = f32x2{cfg[gq + 2 * e], cfg[gq + 2 * e + 1]}; kg2[e] = f32x2{cfg[64 + gq + 2 * e], cfg[64 + gq + 2 * e + 1]};
            if constexpr (G_TWO) kg1[e] = f32x2{cfg[32 + gq + 2 * e], cfg[32 + gq + 2 * e + 1]};
        }
        // v0*k0 (+ v1*k1) + k2, LeakyReLU on the activation operand; pairs -> packed f32 math
        auto xform2 = [&](const Vec16<T>& v0, const Vec16<T>& v1, const f32x2* k0, const f32x2* k1, const f32x2* k2, bool two, float slope)
            __attribute__((always_inline)) {
            Vec16<T> o;
#pragma unroll
            for (int e = 0; e < NE / 2; ++e) {
                const f32x2 x0 = {v0.get(2 * e), v0.get(2 * e + 1)};
                f32x2 z;
                if (two) {
                    const f32x2 x1 = {v1.get(2 * e), v1.get(2 * e + 1)};
                    z = x0 * k0[e] + (x1 * k1[e] + k2[e]);
                } else {
                    z = x0 * k0[e] + k2[e];
                    const f32x2 zs = z * slope;
                    z.x = fmaxf(z.x, zs.x); z.y = fmaxf(z.y, zs.y);
                }
                o.set(2 * e, z.x); o.set(2 * e + 1, z.y);
            }
            return o;
        };
        auto issue_tile = [&](int tile) __attribute__((always_inline)) {
            int b0, y0, x0; tile_origin(tile, b0, y0, x0);
            const int sbase = ((b0 * Hs + y0) * Ws + x0) * CA;
#pragma unroll
            for (int u = 0; u < SIT; ++u) {
                const uint32_t g = (b0 + (sloff[u] >> 20)) < a.B ? (uint32_t)(sbase + srel[u]) * (uint32_t)sizeof(T) : 0u;
                ps0[u] = *reinterpret_cast<const Vec16<T>*>(reinterpret_cast<const char*>(a.s0) + g);
                if constexpr (S_TWO) ps1[u] = *reinterpret_cast<const Vec16<T>*>(reinterpret_cast<const char*>(a.s1) + g);
            }
            const int gbase = ((b0 * Hg + 2 * y0 - 1) * Wg + 2 * x0 - 1) * CB + bc0;
            const int tmask = (y0 == 0 ? 1 << 13 : 0) | (x0 == 0 ? 1 << 14 : 0), nb = a.B - b0;   // uniform per tile
            int2 e[MAXG];
#pragma unroll
            for (int u = 0; u < MAXG; ++u) e[u] = gtab[pt + u * NP];
#pragma unroll
            for (int u = 0; u < MAXG; ++u) {
                const bool ok = ((e[u].y & tmask) == 0) & ((e[u].y >> 15) < nb);
                gmeta[u] = ok ? e[u].y : (e[u].y | (1 << 31));
                const uint32_t g = ok ? (uint32_t)(gbase + e[u].x) * (uint32_t)sizeof(T) : 0u;
                pg0[u] = *reinterpret_cast<const Vec16<T>*>(reinterpret_cast<const char*>(a.g0) + g);
                if constexpr (G_TWO) pg1[u] = *reinterpret_cast<const Vec16<T>*>(reinterpret_cast<const char*>(a.g1) + g);
            }
        };
        if (nt > 0) issue_tile(t_begin);
        for (int i = 0; i <= nt; ++i) {
            if (i < nt) {
                const int tile = t_begin + i;
                char* stile = stile0 + (i & 1) * SBUF;
                char* gtile = gtile0 + (i & 1) * GBUF;
                int b0, y0, x0; tile_origin(tile, b0, y0, x0);
                // transform + store this tile's chunks and, as each register pair becomes free, request the same chunk of the next tile
                const bool nh = i + 1 < nt;
                int nb0 = b0, ny0 = y0, nx0 = x0;
                if (nh) tile_origin(tile + 1, nb0, ny0, nx0);
                const int nsbase = ((nb0 * Hs + ny0) * Ws + nx0) * CA;
                const int ngbase = ((nb0 * Hg + 2 * ny0 - 1) * Wg + 2 * nx0 - 1) * CB + bc0;
                const int ntmask = (ny0 == 0 ? 1 << 13 : 0) | (nx0 == 0 ? 1 << 14 : 0), nnb = a.B - nb0;
#pragma unroll
                for (int u = 0; u < SIT; ++u) {
                    Vec16<T> o = xform2(ps0[u], ps1[S_TWO ? u : 0], ks0, ks1, ks2, S_TWO, a.sslope);
                    if ((b0 + (sloff[u] >> 20)) >= a.B) o = zero_vec16<T>();
                    *reinterpret_cast<Vec16<T>*>(stile + (sloff[u] & 0xfffff)) = o;
                    const uint32_t g = (nh & ((nb0 + (sloff[u] >> 20)) < a.B)) ? (uint32_t)(nsbase + srel[u]) * (uint32_t)sizeof(T) : 0u;
                    ps0[u] = *reinterpret_cast<const Vec16<T>*>(reinterpret_cast<const char*>(a.s0) + g);
                    if constexpr (S_TWO) ps1[u] = *reinterpret_cast<const Vec16<T>*>(reinterpret_cast<const char*>(a.s1) + g);
                }
                WSTAMP(3)
                int2 e[MAXG];
#pragma unroll
                for (int u = 0; u < MAXG; ++u) e[u] = gtab[pt + u * NP];
#pragma unroll
                for (int u = 0; u < MAXG; ++u) {
                    Vec16<T> o = xform2(pg0[u], pg1[G_TWO ? u : 0], kg0, kg1, kg2, G_TWO, a.gslope);
                    if (gmeta[u] < 0) o = zero_vec16<T>();
                    if (pt + u * NP < npix * GCH) *reinterpret_cast<Vec16<T>*>(gtile + ((gmeta[u] & 0x1fff) << 4)) = o;
                    const bool ok = nh & ((e[u].y & ntmask) == 0) & ((e[u].y >> 15) < nnb);
                    gmeta[u] = ok ? e[u].y : (e[u].y | (1 << 31));
                    const uint32_t g = ok ? (uint32_t)(ngbase + e[u].x) * (uint32_t)sizeof(T) : 0u;
                    pg0[u] = *reinterpret_cast<const Vec16<T>*>(reinterpret_cast<const char*>(a.g0) + g);
                    if constexpr (G_TWO) pg1[u] = *reinterpret_cast<const Vec16<T>*>(reinterpret_cast<const char*>(a.g1) + g);
                }
            }
            WSTAMP(1)
            deep::barrier_lds();             // tile i published / tile i-1 consumed (raw: the next tile's global loads stay in flight)
            WSTAMP(2)
        }
    } else {
        // =========================== consumers ===========================
        // 8 waves = 4 blocks of 32 low-res-side channels x 2 tap groups; a wave owns taps ts, ts + 2, ... (no cross-wave sum)
        const int wa = wave & 3, ts = wave >> 2;
        const int g4 = lane >> 4, li = lane & 15, q = li >> 2, p = li & 3;
        f32x16 acc[NTW];
#pragma unroll
        for (int t = 0; t < NTW; ++t)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
        const int colA = (wa * 32 + 16 * (g4 & 1) + 4 * p) * 2, colB = (16 * (g4 & 1) + 4 * p) * 2;
        for (int i = 0; i <= nt; ++i) {
            if (i > 0) {
                const char* stile = stile0 + ((i - 1) & 1) * SBUF;
                const char* gtile = gtile0 + ((i - 1) & 1) * GBUF;
#pragma unroll 2
                for (int ks = 0; ks < WG_KP / 16; ++ks) {
                    const int k0 = ks * 16 + 8 * (g4 >> 1) + q, k1 = k0 + 4;
                    Frag<T> af = frag_tr16<T>(stile + k0 * SPITCH + colA, stile + k1 * SPITCH + colA);
                    const int gb0 = ((k0 >> (a.lth + a.ltw)) * PH + 2 * ((k0 >> a.ltw) & (th - 1))) * PW + 2 * (k0 & (tw - 1));
                    const int gb1 = ((k1 >> (a.lth + a.ltw)) * PH + 2 * ((k1 >> a.ltw) & (th - 1))) * PW + 2 * (k1 & (tw - 1));
#pragma unroll
                    for (int ti = 0; ti < NTW; ++ti) {
                        const int t = ts + ti * TS;
                        if (t < 9) {   // wave-uniform
                            const int ky = (t * 11) >> 5, kx = t - 3 * ky;
                            const int toff = ky * PW + kx;
                            Frag<T> bf = frag_tr16<T>(gtile + (gb0 + toff) * GPITCH + colB, gtile + (gb1 + toff) * GPITCH + colB);
                            mma(acc[ti], af, bf);
                        }
                    }
                }
            }
            WSTAMP(1)
            deep::barrier_lds();
            WSTAMP(2)
        }
        // partial slab: rows = low-res-side channel (a), lanes = high-res-side channel (b)
        const size_t slab_id = blockIdx.x;
#pragma unroll
        for (int ti = 0; ti < NTW; ++ti) {
            const int t = ts + ti * TS;
            if (t < 9) {
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int ca = a0 + wa * 32 + acc_row(i, lane), cb = bc0 + r;
                    a.slab[((slab_id * 9 + t) * CA + ca) * CB + cb] = acc[ti][i];
                }
            }
        }
    }
#ifdef VAE_PHASE_STAMPS
    if (a.dbg && lane == 0) { dst_[5] = clock64() - dst_entry_; for (int k_ = 0; k_ < 8; ++k_) a.dbg[((size_t)(blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z)) * 16 + wave) * 8 + k_] = dst_[k_]; }
#endif
#undef WSTAMP
}
