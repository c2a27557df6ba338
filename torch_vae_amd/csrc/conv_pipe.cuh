// Persistent, software-pipelined versions of down_kernel / up_kernel (conv_mfma.cuh) for gfx950.
//
// Same math and operands; what changes is how the bytes move:
//   * a fixed grid of workgroups walks (M tile, N tile, channel chunk) work items;
//   * the NEXT item's input patch (and, for the backward epilogue, the NEXT tile's y_out rows) is
//     loaded into registers right after the barrier that publishes the current patch, so the HBM
//     round trip overlaps the MFMAs and the epilogue of the current item instead of stalling the wave
//     (rocprof round 1: the one-tile-per-workgroup kernels sat ~70 % of wave-cycles in s_waitcnt);
//   * the epilogue goes through a wave-private LDS tile [pixel][channel]: accumulators are
//     written / combined in place (2-byte LDS cells), then streamed out as whole 16-byte chunks,
//     so every global store (and every y_out load) is a coalesced 16 B per lane;
//   * BatchNorm statistics stay in registers across all tiles of a workgroup: one f64 atomic per
//     channel per workgroup at the very end;
//   * three tile organisations: four waves in a row over a 128-pixel workgroup tile (two barriers per item),
//     wave-independent 32-pixel tiles (WV: no barrier in the loop), a 2x2 wave grid over the 128-pixel x 128-channel
//     tile (LAY = 1: halves the weight-fragment traffic of the deep layers);
//   * what the ISA and the counters taught: the epilogue kind is a template parameter, staging
//     coefficients live in registers, validity comes from flag bits of an LDS geometry table, transforms and statistics
//     use packed f32 math, addresses are 32-bit byte offsets, and weight fragments are requested BEFORE the next
//     item's prefetch burst because vmcnt retires in order.
#pragma once
#include "conv_mfma.cuh"
#include <type_traits>

struct TileGeo { int b0, y0, x0, n0; };

// uniform base + 32-bit BYTE offset: lets the compiler use the scalar-base addressing mode of global_load/store
// (no 64-bit vector address arithmetic); the launcher guarantees every tensor is < 4 GiB
template <typename P> __device__ __forceinline__ P* at_bytes(P* base, uint32_t byte_off) {
    return reinterpret_cast<P*>(const_cast<char*>(reinterpret_cast<const char*>(base)) + byte_off);
}

// round a pair the way it is stored (one packed conversion for bf16) and return the stored values as f32
template <typename T> __device__ __forceinline__ f32x2 round_pair(f32x2 v, T& o0, T& o1);
template <> __device__ __forceinline__ f32x2 round_pair<float>(f32x2 v, float& o0, float& o1) { o0 = v.x; o1 = v.y; return v; }
template <> __device__ __forceinline__ f32x2 round_pair<bf16>(f32x2 v, bf16& o0, bf16& o1) {
    const bf16x2 b = __builtin_convertvector(v, bf16x2);
    o0 = b.x; o1 = b.y;
    return __builtin_convertvector(b, f32x2);
}
// Epilogue of one accumulator pair (two pixels of one channel) against its two LDS cells; statistics are
// accumulated as packed pairs (v_pk_add_f32 / v_pk_fma_f32).  Backward: s2 collects sum dz*y; the caller
// finishes sum dz*xhat = invstd * s2 + xm * s1.
template <typename T, int EPI, bool CHECKED>
__device__ __forceinline__ void epi_pair(float a0, float a1, T* c0, T* c1, bool ok0, bool ok1, float sc, float sh, float oslope,
                                         f32x2& s1, f32x2& s2) {
    f32x2 av = {a0, a1};
    if constexpr (EPI == EPI_FWD) {
        f32x2 v = round_pair<T>(av, *c0, *c1);
        if constexpr (CHECKED) { v.x = ok0 ? v.x : 0.f; v.y = ok1 ? v.y : 0.f; }
        s1 += v; s2 += v * v;
    } else if constexpr (EPI == EPI_BWD) {
        const f32x2 y = {tofloat(*c0), tofloat(*c1)};
        const f32x2 z = y * sc + sh;
        f32x2 g = av;
        g.x = z.x > 0.f ? g.x : g.x * oslope; g.y = z.y > 0.f ? g.y : g.y * oslope;
        const f32x2 dz = round_pair<T>(g, *c0, *c1);
        s1 += dz; s2 += dz * y;
    } else {
        (void)round_pair<T>(av, *c0, *c1);
    }
}

template <> __device__ __forceinline__ f32x2 round_pair<f16>(f32x2 v, f16& o0, f16& o1) {
    const f16x2 b = __builtin_convertvector(v, f16x2);
    o0 = b.x; o1 = b.y;
    return __builtin_convertvector(b, f32x2);
}
template <typename T>
__device__ __forceinline__ TileGeo decode_pair(const ConvArgs<T>& a, int pi, int ntiles_n, int nch_out) {
    int mt = pi / ntiles_n; const int nt = pi - mt * ntiles_n;
    if (a.rev) mt = a.n_mt - 1 - mt;   // reversed walk over the M tiles (the N tile of a workgroup stays)
    const int bt = fastdiv(mt, a.m_txy), trem = mt - bt * a.tiles_x * a.tiles_y, ty = fastdiv(trem, a.m_tx), tx = trem - ty * a.tiles_x;
    TileGeo g; g.b0 = bt << a.lTB; g.y0 = ty << a.lth; g.x0 = tx << a.ltw; g.n0 = nt * nch_out;
    return g;
}

// ---------------------------------------------------------------------------
// WV = wave-independent mode: every wave owns a 32-pixel M tile with its own patch / out tile in LDS and walks
// its own sequence of tiles; no workgroup barrier inside the persistent loop (LDS traffic of one wave is
// processed in order), so the 8 waves of a CU sit in different phases and cover each other's stalls.
// EPI is a template parameter (the per-element epilogue is straight-line code); forward launches stage one source
// tensor with BatchNorm+LeakyReLU, backward launches (EPI_BWD / EPI_PLAIN) stage the two-source gradient operand.
// LAY = 1 (workgroup tiles only, even NT): the 4 waves form a 2x2 grid over the 128-pixel x 32*NT-channel tile,
// each computing 64 pixels x 16*NT channels, so every weight fragment a wave loads feeds two MFMAs - half the
// weight traffic through the vector-memory pipe, which is what bounds the deep layers (wide N, few pixels).
// LAY = 2: the same tile under EIGHT waves (2 x 4 grid, 64 pixels x 8*NT channels each, 512 threads): two waves per
// SIMD, so one wave's staging / epilogue VALU and its waits run under the other's MFMAs (the SQ counters of the
// 4-wave layout: matrix pipe 28 %, VALU 28 %, waiting 38 % of the cycles of the single wave per SIMD).
template <typename T, int NT, int EPI, bool WV, int LAY = 0>
__global__ __launch_bounds__((LAY >= 2 ? 512 : 256), ((NT > 1 || LAY >= 2) ? 1 : 2)) void down2_kernel(ConvArgs<T> a, int n_pairs, int ntiles_n) {
    constexpr bool TWO_SRC = EPI != EPI_FWD;
    constexpr bool W22 = LAY >= 1;                       // waves form a WM x WN grid over the workgroup tile (wave = wm + WM * wn)
    // LAY = 3: eight waves as 4 x 2 (32 pixels x 32 channels each) for 64-channel tiles (NT = 2)
    constexpr int NWV = LAY >= 2 ? 8 : 4, NTHR = 64 * NWV, WN = LAY == 2 ? 4 : (LAY == 1 || LAY == 3 ? 2 : 1), WM = W22 ? NWV / WN : 1;
    static_assert(!W22 || (!WV && NT % WN == 0), "wave-grid layouts: workgroup tiles, NT a multiple of the grid's N width");
    constexpr int MTW = W22 ? 4 / WM : 1, NTW = NT / WN, OROWS = 32 * MTW;   // per wave: M sub-tiles, N sub-tiles, out-tile rows
    constexpr int CK = 64 / sizeof(T), KS = CK / 16, E16 = 16 / sizeof(T), MAXI = LAY >= 2 ? 5 : 10;
    constexpr int OROW = 32 * NTW * sizeof(T), OPITCH = OROW + 16, OCH = OROW / 16;  // out-tile row bytes / chunks
    constexpr int OPL = OROWS * OCH / 64;                                            // out chunks per lane
    extern __shared__ __attribute__((aligned(16))) char smem[];
#ifdef VAE_PHASE_STAMPS
    const long long t_entry = clock64();
#endif
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), r = lane & 31, h = lane >> 5;
    const int stid = WV ? lane : tid, wv0 = WV ? 0 : wave;   // staging thread index; tile-local wave index
    // virtual workgroup id: workgroups are dealt to the 8 XCDs round-robin (XCD = blockIdx % 8); the N tiles of one M tile
    // and neighbouring M tiles are consecutive ids, so each XCD takes a contiguous range of them and the input patch an
    // M tile's N tiles share is fetched into ONE L2 instead of two to four (a.xcd = gridDim/8, 0: identity)
    const int vb = a.xcd ? (int)(blockIdx.x & 7) * a.xcd + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
    const int wm = W22 ? (wave % WM) : wv0, wn = W22 ? (wave / WM) : 0;   // wave coordinates in the workgroup tile
    const int mrow0 = wm * OROWS;                                         // first tile pixel of this wave
    constexpr int SSTR = WV ? 64 : NTHR;
    const int th = 1 << a.lth, tw = 1 << a.ltw, TB = 1 << a.lTB;
    const int PH = 2 * th + 1, PW = 2 * tw + 1, PP = PH * PW, npix = TB * PP, nitems = npix * 4;
    const int Hin = 2 * a.Hs, Win = 2 * a.Ws, Cin = a.Cin, Cout = a.Cout, NCH = Cin / CK;
    float* cf = reinterpret_cast<float*>(smem);
    char* patch0 = smem + ((3 * Cin * 4 + 15) & ~15);
    char* patch = patch0 + (WV ? wave * npix * PATCH_PITCH : 0);
    char* otile = patch0 + (WV ? 4 : 1) * npix * PATCH_PITCH;                 // [4 waves][32 px][OPITCH]
    float* red = reinterpret_cast<float*>(otile + NWV * OROWS * OPITCH);
    char* mytile = otile + wave * OROWS * OPITCH;
    // per-item staging table (tile-independent): {relative global element offset, LDS offset/16 | top<<13 | left<<14 | img<<15}
    int2* itab = reinterpret_cast<int2*>(red + 4 * NT * 32 * 2);
    // (padded to MAXI*SSTR entries; padding entries carry image 0xffff, which never passes the batch test)
    for (int it = tid; it < max(nitems, MAXI * SSTR); it += NTHR) {
        const int pix = it >> 2, q = it & 3;
        const int img = fastdiv(pix, a.m_pp), rem = pix - img * PP, py = fastdiv(rem, a.m_pw), px = rem - py * PW;
        itab[it] = it < nitems ? make_int2(((img * Hin + py) * Win + px) * Cin + q * E16,
                                           ((pix * PATCH_PITCH + q * 16) >> 4) | ((py == 0) << 13) | ((px == 0) << 14) | (img << 15))
                               : make_int2(0, 0xffff << 15);
    }

    int pbase[MTW];
#pragma unroll
    for (int mt = 0; mt < MTW; ++mt) {
        const int R = mrow0 + mt * 32 + r;
        pbase[mt] = ((R >> (a.lth + a.ltw)) * PH + 2 * ((R >> a.ltw) & (th - 1))) * PW + 2 * (R & (tw - 1));
    }

    // one N tile per workgroup for its whole life: epilogue coefficients live in registers; the forward
    // accumulators start from the conv bias
    float ebv[NTW], esc[NTW], esh[NTW], eis[NTW], exm[NTW];
    {
        const int n0w = (vb % ntiles_n) * 32 * NT + wn * NTW * 32;
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt) {
            const int n = n0w + nt * 32 + r;
            ebv[nt] = (EPI == EPI_FWD && a.bias) ? a.bias[n] : 0.f;
            esc[nt] = esh[nt] = eis[nt] = exm[nt] = 0.f;
            if constexpr (EPI == EPI_BWD) {
                esc[nt] = a.ocoef[LC_SC * Cout + n]; esh[nt] = a.ocoef[LC_SH * Cout + n];
                eis[nt] = a.ocoef[LC_INVSTD * Cout + n]; exm[nt] = a.ocoef[LC_XM * Cout + n];
            }
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

    // a thread always stages the same 16-byte quarter of a pixel (stid & 3): its per-channel coefficients live in
    // registers, reloaded only when the channel chunk changes
    constexpr int NE = Vec16<T>::N;
    float k0[NE], k1[TWO_SRC ? NE : 1], k2[NE];
    auto load_coefs = [&](int c0) __attribute__((always_inline)) {
        const int cb = c0 + (stid & 3) * E16;
#pragma unroll
        for (int e = 0; e < NE; ++e) {
            k0[e] = cf[cb + e]; k2[e] = cf[2 * Cin + cb + e];
            if constexpr (TWO_SRC) k1[e] = cf[Cin + cb + e];
        }
    };
    auto xform = [&](const Vec16<T>& v0, const Vec16<T>& v1) __attribute__((always_inline)) {
        Vec16<T> o;
#pragma unroll
        for (int e = 0; e < NE; e += 2) {   // pairs: packed f32 multiply-adds
            const f32x2 x0 = {v0.get(e), v0.get(e + 1)}, c0 = {k0[e], k0[e + 1]}, c2 = {k2[e], k2[e + 1]};
            f32x2 z;
            if constexpr (TWO_SRC) {
                const f32x2 x1 = {v1.get(e), v1.get(e + 1)}, c1 = {k1[e], k1[e + 1]};
                z = x0 * c0 + (x1 * c1 + c2);   // two packed fmas
            } else {
                z = x0 * c0 + c2;
                const f32x2 zs = z * a.slope;
                z.x = fmaxf(z.x, zs.x); z.y = fmaxf(z.y, zs.y);
            }
            o.set(e, z.x); o.set(e + 1, z.y);
        }
        return o;
    };
    Vec16<T> pre0[MAXI], pre1[TWO_SRC ? MAXI : 1];
    decltype(Vec16<T>::v) prey[OPL];   // raw vectors: keeps the prefetch in VGPRs (a struct array was demoted to scratch)

    // item -> (LDS offset, validity, global offset) from the LDS table; recomputed where needed instead of kept in
    // registers.  Offsets are 32-bit element indices (the launcher checks the tensors are < 2^31 elements), so the
    // loads use the scalar-base + 32-bit-offset form and validity is a select, not a branch.
    auto tile_base = [&](const TileGeo& g, int c0) __attribute__((always_inline)) {
        return ((g.b0 * Hin + 2 * g.y0 - 1) * Win + 2 * g.x0 - 1) * Cin + c0;
    };
    auto item_ok = [&](const TileGeo& g, int it, int ey) __attribute__((always_inline)) {
        // the halo row/column (py==0 / px==0: flag bits 13/14) falls outside the image only for tiles on the top / left border
        const int tmask = (g.y0 == 0 ? 1 << 13 : 0) | (g.x0 == 0 ? 1 << 14 : 0);   // uniform per tile
        return ((ey & tmask) == 0) & ((ey >> 15) < a.B - g.b0);
    };
    auto issue = [&](const TileGeo& g, int c0) __attribute__((always_inline)) {
        const int base = tile_base(g, c0);
        int2 e[MAXI];
#pragma unroll
        for (int u = 0; u < MAXI; ++u) { const int it = stid + u * SSTR; e[u] = itab[it]; }
#pragma unroll
        for (int u = 0; u < MAXI; ++u) {
            const uint32_t gi = item_ok(g, stid + u * SSTR, e[u].y) ? (uint32_t)(base + e[u].x) : 0u;
            pre0[u] = *reinterpret_cast<const Vec16<T>*>(at_bytes(a.src0, gi * (uint32_t)sizeof(T)));
            if constexpr (TWO_SRC) pre1[u] = *reinterpret_cast<const Vec16<T>*>(at_bytes(a.src1, gi * (uint32_t)sizeof(T)));
        }
    };
    // Transform + store the staged chunks of item g and, chunk by chunk as its registers become free, request the same
    // chunk of the NEXT item gn (one table entry serves both: the table is tile-independent).  Starting the next burst
    // here instead of after the barrier lengthens its flight time by the whole transform phase.
    auto write_patch = [&](const TileGeo& g, int c0, const TileGeo& gn, int cn0, bool nhave_) __attribute__((always_inline)) {
        int2 e[MAXI];
#pragma unroll
        for (int u = 0; u < MAXI; ++u) { const int it = stid + u * SSTR; e[u] = itab[it]; }
        const int nbase = tile_base(gn, cn0);
#pragma unroll
        for (int u = 0; u < MAXI; ++u) {
            const int it = stid + u * SSTR;
            Vec16<T> o = xform(pre0[u], pre1[TWO_SRC ? u : 0]);
            const bool ok_ = item_ok(g, it, e[u].y);
            if (!ok_) o = zero_vec16<T>();
            if (it < nitems) *reinterpret_cast<Vec16<T>*>(patch + ((e[u].y & 0x1fff) << 4)) = o;
            if (a.stage_out && ok_ && g.n0 == 0 && !(e[u].y & (3 << 13)) && it < nitems)   // owned (non-halo) chunk: materialise it
                *reinterpret_cast<Vec16<T>*>(at_bytes(a.stage_out, (uint32_t)(tile_base(g, c0) + e[u].x) * (uint32_t)sizeof(T))) = o;
            const uint32_t gi = (nhave_ & item_ok(gn, it, e[u].y)) ? (uint32_t)(nbase + e[u].x) : 0u;
            pre0[u] = *reinterpret_cast<const Vec16<T>*>(at_bytes(a.src0, gi * (uint32_t)sizeof(T)));
            if constexpr (TWO_SRC) pre1[u] = *reinterpret_cast<const Vec16<T>*>(at_bytes(a.src1, gi * (uint32_t)sizeof(T)));
        }
        // patches larger than MAXI*SSTR chunks (tiny spatial sizes, many images per tile): synchronous tail
        for (int it = stid + MAXI * SSTR; it < nitems; it += SSTR) {
            const int2 e = itab[it];
            const bool ok = item_ok(g, it, e.y);
            Vec16<T> v = zero_vec16<T>();
            if (ok) v = load_transform16<T>(a.src0, a.src1, TWO_SRC, (size_t)(tile_base(g, c0) + e.x), cf, Cin, c0 + (it & 3) * E16, a.slope);
            *reinterpret_cast<Vec16<T>*>(patch + ((e.y & 0x1fff) << 4)) = v;
            if (a.stage_out && ok && g.n0 == 0 && !(e.y & (3 << 13)))
                *reinterpret_cast<Vec16<T>*>(at_bytes(a.stage_out, (uint32_t)(tile_base(g, c0) + e.x) * (uint32_t)sizeof(T))) = v;
        }
    };
    // out-tile chunk u of this lane (wave-local chunk id = lane + 64u): the tile-independent part of its address is
    // computed once: orel = element offset relative to the tile origin, opk = LDS offset | image-in-tile << 20
    int orel[OPL], opk[OPL];
#pragma unroll
    for (int u = 0; u < OPL; ++u) {
        const int id = lane + 64 * u, row = id / OCH, qq = id - row * OCH, RR = mrow0 + row;
        const int img = RR >> (a.lth + a.ltw), ty = (RR >> a.ltw) & (th - 1), tx = RR & (tw - 1);
        orel[u] = ((img * a.Hs + ty) * a.Ws + tx) * Cout + wn * NTW * 32 + qq * E16;
        opk[u] = (row * OPITCH + qq * 16) | (img << 20);
    }
    // global element offset of out-tile chunk u, or -1 (image beyond the batch)
    auto out_chunk_addr = [&](const TileGeo& g, int u, int& loff) __attribute__((always_inline)) -> int {
        loff = opk[u] & 0xfffff;
        const int base = ((g.b0 * a.Hs + g.y0) * a.Ws + g.x0) * Cout + g.n0;
        return (g.b0 + (opk[u] >> 20)) < a.B ? base + orel[u] : -1;
    };
    auto issue_y = [&](const TileGeo& g) __attribute__((always_inline)) {
#pragma unroll
        for (int u = 0; u < OPL; ++u) {
            int loff; const int gi = out_chunk_addr(g, u, loff);
            prey[u] = *reinterpret_cast<const decltype(Vec16<T>::v)*>(at_bytes(a.yout, (uint32_t)(gi < 0 ? 0 : gi) * (uint32_t)sizeof(T)));
        }
    };

    // WV: the workgroup keeps its N tile (vb % ntiles_n); its waves take adjacent M tiles
    int pi = WV ? ((vb / ntiles_n) * 4 + wave) * ntiles_n + vb % ntiles_n : vb, chunk = 0;
    const int pstride = WV ? 4 * gridDim.x : gridDim.x;
    if (a.fuse.mode != BNF_NONE) {   // finalise the input layer's BatchNorm here (workgroup 0 also records it)
        for (int i = tid; i < Cin; i += NTHR) bn_fused_channel(a.fuse, i, blockIdx.x == 0, cf[i], cf[Cin + i], cf[2 * Cin + i]);
    } else {
        for (int i = tid; i < 3 * Cin; i += NTHR) cf[i] = a.coef[i];
    }
    __syncthreads();                                       // staging table / coefficients published
    load_coefs(0);
    bool have = pi < n_pairs;
    TileGeo cur = decode_pair(a, have ? pi : 0, ntiles_n, 32 * NT);
    if (have) { issue(cur, 0); if (EPI == EPI_BWD) issue_y(cur); }
#ifdef VAE_PHASE_STAMPS
    long long tph[8] = {0, 0, 0, 0, 0, 0, 0, 0}; long long t0 = clock64(); tph[6] = t0 - t_entry;
#endif
#ifdef VAE_PHASE_STAMPS   // diagnostic build (make STAMPS=1): the stamps split the loop body into scheduling regions
#define STAMP(k) { if (a.dbg) { __builtin_amdgcn_sched_barrier(0); long long t1 = clock64(); tph[k] += t1 - t0; t0 = t1; __builtin_amdgcn_sched_barrier(0); } }
#else
#define STAMP(k)
#endif
    while (have) {
        if (WV) asm volatile("" ::: "memory"); else __syncthreads();   // (A) previous item fully consumed
        STAMP(0)
        // Weights straight from L1/L2 as the B operand.  vmcnt retires in order, so they are requested FIRST, ahead of
        // the next item's prefetch burst (issued below): a weight load queued behind the burst could not be consumed
        // before the whole burst - HBM latency - had returned.  All nine taps when the registers allow it.
        const int c0 = chunk * CK;
        constexpr int FRAG_REGS = 16 / 4 * (sizeof(T) == 2 ? 1 : 2);
        constexpr int DEPTH = NT == 1 ? 4 : ((9 * KS * NTW * FRAG_REGS <= 160) ? ((W22 && EPI != EPI_BWD) ? 6 : 9) : 4);
        Frag<T> bq[DEPTH][KS][NTW];
        auto load_b = [&](int t, int slot) __attribute__((always_inline)) {
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const uint32_t kg = (uint32_t)t * (Cin >> 3) + ((c0 + ks * 16) >> 3) + h;
#pragma unroll
                for (int nt = 0; nt < NTW; ++nt) bq[slot][ks][nt] = load_frag(at_bytes(a.wp, (kg * Cout + cur.n0 + (wn * NTW + nt) * 32 + r) * (uint32_t)(8 * sizeof(T))));
            }
        };
#pragma unroll
        for (int dd = 0; dd < DEPTH; ++dd) load_b(dd, dd);
        if (NCH > 1) load_coefs(chunk * CK);
        int npi = pi, nchunk = chunk + 1;
        TileGeo nxt = cur;
        if (nchunk == NCH) { nchunk = 0; npi += pstride; if (npi < n_pairs) nxt = decode_pair(a, npi, ntiles_n, 32 * NT); }
        const bool nhave = npi < n_pairs;
        write_patch(cur, chunk * CK, nxt, nchunk * CK, nhave);   // ... and the next item's loads, in flight from here on
        if (EPI == EPI_BWD && chunk == 0) {
#pragma unroll
            for (int u = 0; u < OPL; ++u) {
                int loff; (void)out_chunk_addr(cur, u, loff);
                *reinterpret_cast<decltype(Vec16<T>::v)*>(mytile + loff) = prey[u];
            }
        }
        STAMP(1)
        if (WV) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); else __syncthreads();   // (B) patch published
        STAMP(2)

        STAMP(3)
        {
            // A fragments one (tap, k-step) ahead of the matrix pipe
            auto load_a = [&](int step, int mt) __attribute__((always_inline)) {
                const int t = step / KS, ks = step % KS;
                return load_frag(reinterpret_cast<const T*>(patch + (pbase[mt] + (t / 3) * PW + (t % 3)) * PATCH_PITCH + ks * 32) + h * 8);
            };
            Frag<T> af[MTW], afn[MTW];
#pragma unroll
            for (int mt = 0; mt < MTW; ++mt) af[mt] = load_a(0, mt);
#pragma unroll
            for (int t = 0; t < 9; ++t) {
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    const int step = t * KS + ks;
#pragma unroll
                    for (int mt = 0; mt < MTW; ++mt) { afn[mt] = af[mt]; if (step + 1 < 9 * KS) afn[mt] = load_a(step + 1, mt); }
#pragma unroll
                    for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
                        for (int mt = 0; mt < MTW; ++mt) mma(acc[mt][nt], af[mt], bq[t % DEPTH][ks][nt]);
#pragma unroll
                    for (int mt = 0; mt < MTW; ++mt) af[mt] = afn[mt];
                }
                if (t + DEPTH < 9 && !(a.two_src & 2)) load_b(t + DEPTH, t % DEPTH);
            }
        }

        STAMP(4)
        if (chunk == NCH - 1) {
            if (nhave && nchunk == 0 && EPI == EPI_BWD) issue_y(nxt);
            // ---- epilogue through the wave-private LDS tile.  Rows of images beyond the batch (only possible in the
            // last batch tile) carry acc == 0: they add nothing to the backward statistics, and the forward
            // statistics skip them through the checked variant, chosen once per tile.
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
            if (EPI != EPI_FWD || cur.b0 + TB <= a.B) epi_body(std::false_type{}); else epi_body(std::true_type{});
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // own LDS writes landed (wave-private rows)
#pragma unroll
            for (int u = 0; u < OPL; ++u) {
                int loff; const int gi = out_chunk_addr(cur, u, loff);
                const Vec16<T> v = *reinterpret_cast<const Vec16<T>*>(mytile + loff);
                if (gi >= 0) *reinterpret_cast<Vec16<T>*>(at_bytes(a.out, (uint32_t)gi * (uint32_t)sizeof(T))) = v;
            }
        }
        STAMP(5)
        pi = npi; chunk = nchunk; cur = nxt; have = nhave;
    }
#ifdef VAE_PHASE_STAMPS
    const long long t_loop_end = clock64();
#endif
#undef STAMP

    if constexpr (EPI != EPI_PLAIN) {
        // all (M tile, N tile) pairs of one workgroup may span several N tiles: statistics are kept per
        // pair's channel block only when ntiles_n == 1; otherwise flush per pair (see launcher: grid is
        // arranged so that a workgroup keeps one N tile).
        __syncthreads();
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt) {
            float v1 = s1[nt].x + s1[nt].y, v2 = s2[nt].x + s2[nt].y;
            if constexpr (EPI == EPI_BWD) v2 = eis[nt] * v2 + exm[nt] * v1;   // sum dz*xhat from sum dz*y and sum dz
            v1 += __shfl_xor(v1, 32, 64); v2 += __shfl_xor(v2, 32, 64);
            if (h == 0) { red[((wave * NTW + nt) * 32 + r) * 2] = v1; red[((wave * NTW + nt) * 32 + r) * 2 + 1] = v2; }
        }
        __syncthreads();
        if (tid < NT * 32) {
            float v1 = 0.f, v2 = 0.f;
            if constexpr (W22) {   // channel tid lives in the WM waves of column wn = tid / (32*NTW)
                const int cw = tid / (32 * NTW), cl = tid - cw * 32 * NTW;
#pragma unroll
                for (int m = 0; m < WM; ++m) { v1 += red[(((m + WM * cw) * NTW) * 32 + cl) * 2]; v2 += red[(((m + WM * cw) * NTW) * 32 + cl) * 2 + 1]; }
            } else {
#pragma unroll
                for (int w = 0; w < 4; ++w) { v1 += red[((w * NT) * 32 + tid) * 2]; v2 += red[((w * NT) * 32 + tid) * 2 + 1]; }
            }
            const int n0 = (vb % ntiles_n) * 32 * NT;
            double* st_ = a.stat + stat_rep() * 2 * Cout;
            unsafeAtomicAdd(&st_[n0 + tid], (double)v1);
            unsafeAtomicAdd(&st_[Cout + n0 + tid], (double)v2);
        }
    }
#ifdef VAE_PHASE_STAMPS
    if (a.dbg && lane == 0) {
        tph[7] = clock64() - t_loop_end;
        for (int k = 0; k < 8; ++k) a.dbg[(blockIdx.x * NWV + wave) * 8 + k] = tph[k];
    }
#endif
}

// ---------------------------------------------------------------------------
template <typename T, int NT, int EPI, bool WV>
__global__ __launch_bounds__(256, ((NT > 1 || (sizeof(T) == 4 && EPI == EPI_BWD)) ? 1 : 2)) void up2_kernel(ConvArgs<T> a, int n_pairs, int ntiles_n) {
    constexpr bool TWO_SRC = EPI != EPI_FWD;
    constexpr int CK = 64 / sizeof(T), KS = CK / 16, E16 = 16 / sizeof(T), MAXI = 3;
    constexpr int OROW = 32 * NT * sizeof(T), OPITCH = OROW + 16, OCH = OROW / 16;
    constexpr int OPL = OCH;                          // 64 output pixels per wave per round (32 base px x 2 x-parities)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), r = lane & 31, h = lane >> 5;
    const int stid = WV ? lane : tid, wv0 = WV ? 0 : wave;   // staging thread index; tile-local wave index
    // virtual workgroup id: workgroups are dealt to the 8 XCDs round-robin (XCD = blockIdx % 8); the N tiles of one M tile
    // and neighbouring M tiles are consecutive ids, so each XCD takes a contiguous range of them and the input patch an
    // M tile's N tiles share is fetched into ONE L2 instead of two to four (a.xcd = gridDim/8, 0: identity)
    const int vb = a.xcd ? (int)(blockIdx.x & 7) * a.xcd + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
    constexpr int SSTR = WV ? 64 : 256;
    const int th = 1 << a.lth, tw = 1 << a.ltw, TB = 1 << a.lTB;
    const int PH = th + 1, PW = tw + 1, PP = PH * PW, npix = TB * PP, nitems = npix * 4;
    const int Hs = a.Hs, Ws = a.Ws, Cin = a.Cin, Cout = a.Cout, NCH = Cin / CK;
    float* cf = reinterpret_cast<float*>(smem);
    char* patch0 = smem + ((3 * Cin * 4 + 15) & ~15);
    char* patch = patch0 + (WV ? wave * npix * PATCH_PITCH : 0);
    char* otile = patch0 + (WV ? 4 : 1) * npix * PATCH_PITCH;                 // [4 waves][64 px][OPITCH]
    float* red = reinterpret_cast<float*>(otile + 256 * OPITCH);
    char* mytile = otile + wave * 64 * OPITCH;
    // per-item staging table: {relative global element offset, LDS offset/16 | bottom-halo<<13 | right-halo<<14 | img<<15}
    int2* itab = reinterpret_cast<int2*>(red + 4 * NT * 32 * 2);
    for (int it = tid; it < max(nitems, MAXI * SSTR); it += 256) {
        const int pix = it >> 2, q = it & 3;
        const int img = fastdiv(pix, a.m_pp), rem = pix - img * PP, py = fastdiv(rem, a.m_pw), px = rem - py * PW;
        itab[it] = it < nitems ? make_int2(((img * Hs + py) * Ws + px) * Cin + q * E16,
                                           ((pix * PATCH_PITCH + q * 16) >> 4) | ((py == th) << 13) | ((px == tw) << 14) | (img << 15))
                               : make_int2(0, 0xffff << 15);
    }

    const int R = wv0 * 32 + r;
    const int pbase = ((R >> (a.lth + a.ltw)) * PH + ((R >> a.ltw) & (th - 1))) * PW + (R & (tw - 1));

    f32x16 acc[4][NT];
    f32x2 s1[NT], s2[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        s1[nt] = f32x2{0.f, 0.f}; s2[nt] = f32x2{0.f, 0.f};
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[c][nt][i] = 0.f;   // (re-initialised with the conv bias below)
    }
    constexpr int NTAP = 9;
    constexpr int tap_cls[NTAP] = {0, 1, 1, 2, 2, 3, 3, 3, 3};
    constexpr int tap_t[NTAP] = {4, 5, 3, 7, 1, 8, 6, 2, 0};
    constexpr int tap_off[NTAP] = {0, 0, 1, 0, 2, 0, 1, 2, 3};  // di*2+dj

    // a thread always stages the same 16-byte quarter of a pixel (stid & 3): its per-channel coefficients live in
    // registers, reloaded only when the channel chunk changes
    // (the backward variant keeps them in LDS: its 4 parity accumulators + two-source prefetch fill the register file)
    constexpr int NE = Vec16<T>::N;
    constexpr bool KREG = !TWO_SRC;
    float k0[KREG ? NE : 1], k2[KREG ? NE : 1];
    int kcb = 0;
    auto load_coefs = [&](int c0) __attribute__((always_inline)) {
        kcb = c0 + (stid & 3) * E16;
        if constexpr (KREG) {
#pragma unroll
            for (int e = 0; e < NE; ++e) { k0[e] = cf[kcb + e]; k2[e] = cf[2 * Cin + kcb + e]; }
        }
    };
    auto xform = [&](const Vec16<T>& v0, const Vec16<T>& v1) __attribute__((always_inline)) {
        Vec16<T> o;
#pragma unroll
        for (int e = 0; e < NE; e += 2) {   // pairs: packed f32 multiply-adds
            const f32x2 x0 = {v0.get(e), v0.get(e + 1)};
            f32x2 z;
            if constexpr (TWO_SRC) {
                const f32x2 x1 = {v1.get(e), v1.get(e + 1)};
                const f32x2 c0 = {cf[kcb + e], cf[kcb + e + 1]}, c1 = {cf[Cin + kcb + e], cf[Cin + kcb + e + 1]}, c2 = {cf[2 * Cin + kcb + e], cf[2 * Cin + kcb + e + 1]};
                z = x0 * c0 + (x1 * c1 + c2);   // two packed fmas
            } else {
                const f32x2 c0 = {k0[e], k0[e + 1]}, c2 = {k2[e], k2[e + 1]};
                z = x0 * c0 + c2;
                const f32x2 zs = z * a.slope;
                z.x = fmaxf(z.x, zs.x); z.y = fmaxf(z.y, zs.y);
            }
            o.set(e, z.x); o.set(e + 1, z.y);
        }
        return o;
    };
    Vec16<T> pre0[MAXI], pre1[TWO_SRC ? MAXI : 1];
    decltype(Vec16<T>::v) prey[OPL];

    auto tile_base = [&](const TileGeo& g, int c0) __attribute__((always_inline)) {
        return ((g.b0 * Hs + g.y0) * Ws + g.x0) * Cin + c0;
    };
    auto item_ok = [&](const TileGeo& g, int it, int ey) __attribute__((always_inline)) {
        // the bottom/right halo (row th, column tw: flag bits 13/14) leaves the image only on the last tile row / column
        const int tmask = (g.y0 + th >= Hs ? 1 << 13 : 0) | (g.x0 + tw >= Ws ? 1 << 14 : 0);   // uniform per tile
        return ((ey & tmask) == 0) & ((ey >> 15) < a.B - g.b0);
    };
    auto issue = [&](const TileGeo& g, int c0) __attribute__((always_inline)) {
        const int base = tile_base(g, c0);
        int2 e[MAXI];
#pragma unroll
        for (int u = 0; u < MAXI; ++u) { const int it = stid + u * SSTR; e[u] = itab[it]; }
#pragma unroll
        for (int u = 0; u < MAXI; ++u) {
            const uint32_t gi = item_ok(g, stid + u * SSTR, e[u].y) ? (uint32_t)(base + e[u].x) : 0u;
            pre0[u] = *reinterpret_cast<const Vec16<T>*>(at_bytes(a.src0, gi * (uint32_t)sizeof(T)));
            if constexpr (TWO_SRC) pre1[u] = *reinterpret_cast<const Vec16<T>*>(at_bytes(a.src1, gi * (uint32_t)sizeof(T)));
        }
    };
    // Transform + store the staged chunks of item g and, chunk by chunk as its registers become free, request the same
    // chunk of the NEXT item gn (one table entry serves both: the table is tile-independent).  Starting the next burst
    // here instead of after the barrier lengthens its flight time by the whole transform phase.
    auto write_patch = [&](const TileGeo& g, int c0, const TileGeo& gn, int cn0, bool nhave_) __attribute__((always_inline)) {
        int2 e[MAXI];
#pragma unroll
        for (int u = 0; u < MAXI; ++u) { const int it = stid + u * SSTR; e[u] = itab[it]; }
        const int nbase = tile_base(gn, cn0);
#pragma unroll
        for (int u = 0; u < MAXI; ++u) {
            const int it = stid + u * SSTR;
            Vec16<T> o = xform(pre0[u], pre1[TWO_SRC ? u : 0]);
            const bool ok_ = item_ok(g, it, e[u].y);
            if (!ok_) o = zero_vec16<T>();
            if (it < nitems) *reinterpret_cast<Vec16<T>*>(patch + ((e[u].y & 0x1fff) << 4)) = o;
            if (a.stage_out && ok_ && g.n0 == 0 && !(e[u].y & (3 << 13)) && it < nitems)   // owned (non-halo) chunk: materialise it
                *reinterpret_cast<Vec16<T>*>(at_bytes(a.stage_out, (uint32_t)(tile_base(g, c0) + e[u].x) * (uint32_t)sizeof(T))) = o;
            const uint32_t gi = (nhave_ & item_ok(gn, it, e[u].y)) ? (uint32_t)(nbase + e[u].x) : 0u;
            pre0[u] = *reinterpret_cast<const Vec16<T>*>(at_bytes(a.src0, gi * (uint32_t)sizeof(T)));
            if constexpr (TWO_SRC) pre1[u] = *reinterpret_cast<const Vec16<T>*>(at_bytes(a.src1, gi * (uint32_t)sizeof(T)));
        }
        for (int it = stid + MAXI * SSTR; it < nitems; it += SSTR) {
            const int2 e = itab[it];
            const bool ok = item_ok(g, it, e.y);
            Vec16<T> v = zero_vec16<T>();
            if (ok) v = load_transform16<T>(a.src0, a.src1, TWO_SRC, (size_t)(tile_base(g, c0) + e.x), cf, Cin, c0 + (it & 3) * E16, a.slope);
            *reinterpret_cast<Vec16<T>*>(patch + ((e.y & 0x1fff) << 4)) = v;
            if (a.stage_out && ok && g.n0 == 0 && !(e.y & (3 << 13)))
                *reinterpret_cast<Vec16<T>*>(at_bytes(a.stage_out, (uint32_t)(tile_base(g, c0) + e.x) * (uint32_t)sizeof(T))) = v;
        }
    };
    // round py: wave-local out pixel o = 2*row + px (row = base pixel 0..31) -> LDS row o, global (2i+py, 2j+px).
    // The tile-independent part of each lane's chunk addresses is computed once (orel, opk as in down2_kernel).
    int orel[OPL], opk[OPL];
#pragma unroll
    for (int u = 0; u < OPL; ++u) {
        const int id = lane + 64 * u, o = id / OCH, qq = id - o * OCH, row = o >> 1, px = o & 1, RR = wv0 * 32 + row;
        const int img = RR >> (a.lth + a.ltw), ty = (RR >> a.ltw) & (th - 1), tx = RR & (tw - 1);
        orel[u] = ((img * 2 * Hs + 2 * ty) * 2 * Ws + 2 * tx + px) * Cout + qq * E16;
        opk[u] = (o * OPITCH + qq * 16) | (img << 20);
    }
    auto out_chunk_addr = [&](const TileGeo& g, int py, int u, int& loff) __attribute__((always_inline)) -> int {
        loff = opk[u] & 0xfffff;
        const int base = ((g.b0 * 2 * Hs + 2 * g.y0 + py) * 2 * Ws + 2 * g.x0) * Cout + g.n0;
        return (g.b0 + (opk[u] >> 20)) < a.B ? base + orel[u] : -1;
    };
    auto issue_y = [&](const TileGeo& g, int py) __attribute__((always_inline)) {
#pragma unroll
        for (int u = 0; u < OPL; ++u) {
            int loff; const int gi = out_chunk_addr(g, py, u, loff);
            prey[u] = *reinterpret_cast<const decltype(Vec16<T>::v)*>(at_bytes(a.yout, (uint32_t)(gi < 0 ? 0 : gi) * (uint32_t)sizeof(T)));
        }
    };

    // one N tile per workgroup for its whole life: epilogue coefficients live in registers
    float ebv[NT], esc[NT], esh[NT], eis[NT], exm[NT];
    {
        const int n0w = (vb % ntiles_n) * 32 * NT;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int n = n0w + nt * 32 + r;
            ebv[nt] = (EPI == EPI_FWD && a.bias) ? a.bias[n] : 0.f;
            esc[nt] = esh[nt] = eis[nt] = exm[nt] = 0.f;
            if constexpr (EPI == EPI_BWD) {
                esc[nt] = a.ocoef[LC_SC * Cout + n]; esh[nt] = a.ocoef[LC_SH * Cout + n];
                eis[nt] = a.ocoef[LC_INVSTD * Cout + n]; exm[nt] = a.ocoef[LC_XM * Cout + n];
            }
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[c][nt][i] = ebv[nt];   // forward accumulators start from the conv bias
        }
    }
    // WV: the workgroup keeps its N tile (vb % ntiles_n); its waves take adjacent M tiles
    int pi = WV ? ((vb / ntiles_n) * 4 + wave) * ntiles_n + vb % ntiles_n : vb, chunk = 0;
    const int pstride = WV ? 4 * gridDim.x : gridDim.x;
    if (a.fuse.mode != BNF_NONE) {   // finalise the input layer's BatchNorm here (workgroup 0 also records it)
        for (int i = tid; i < Cin; i += 256) bn_fused_channel(a.fuse, i, blockIdx.x == 0, cf[i], cf[Cin + i], cf[2 * Cin + i]);
    } else {
        for (int i = tid; i < 3 * Cin; i += 256) cf[i] = a.coef[i];
    }
    __syncthreads();                                       // staging table / coefficients published
    load_coefs(0);
    bool have = pi < n_pairs;
    TileGeo cur = decode_pair(a, have ? pi : 0, ntiles_n, 32 * NT);
    if (have) issue(cur, 0);
#ifdef VAE_PHASE_STAMPS
    long long tph[6] = {0, 0, 0, 0, 0, 0}; long long t0 = clock64();
#endif
#ifdef VAE_PHASE_STAMPS   // diagnostic build (make STAMPS=1): the stamps split the loop body into scheduling regions
#define STAMP(k) { if (a.dbg) { __builtin_amdgcn_sched_barrier(0); long long t1 = clock64(); tph[k] += t1 - t0; t0 = t1; __builtin_amdgcn_sched_barrier(0); } }
#else
#define STAMP(k)
#endif
    while (have) {
        if (WV) asm volatile("" ::: "memory"); else __syncthreads();   // (A)
        STAMP(0)
        // weight fragments of the first DEPTH (k-step, tap) steps are requested before the prefetch burst of the next item
        // (vmcnt retires in order: see down2_kernel); the queue then runs DEPTH steps ahead of the matrix pipe
        const int c0 = chunk * CK;
        constexpr int DEPTH = sizeof(T) == 4 ? (NT == 1 ? 4 : 2) : (NT == 1 ? (EPI == EPI_BWD ? 9 : 10) : 3);
        Frag<T> bq[DEPTH][NT];
        auto load_b = [&](int st_, int slot) __attribute__((always_inline)) {
            const int ks = st_ / NTAP, k = st_ % NTAP;
            const uint32_t kg = (uint32_t)tap_t[k] * (Cin >> 3) + ((c0 + ks * 16) >> 3) + h;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) bq[slot][nt] = load_frag(at_bytes(a.wp, (kg * Cout + cur.n0 + nt * 32 + r) * (uint32_t)(8 * sizeof(T))));
        };
#pragma unroll
        for (int dd = 0; dd < DEPTH; ++dd) if (dd < KS * NTAP) load_b(dd, dd);
        if (NCH > 1) load_coefs(chunk * CK);
        int npi = pi, nchunk = chunk + 1;
        TileGeo nxt = cur;
        if (nchunk == NCH) { nchunk = 0; npi += pstride; if (npi < n_pairs) nxt = decode_pair(a, npi, ntiles_n, 32 * NT); }
        const bool nhave = npi < n_pairs;
        write_patch(cur, chunk * CK, nxt, nchunk * CK, nhave);
        STAMP(1)
        if (WV) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); else __syncthreads();   // (B)
        STAMP(2)
        if (chunk == NCH - 1 && EPI == EPI_BWD) issue_y(cur, 0);   // rows of round 0, hidden behind the MFMAs
        STAMP(3)

#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            Frag<T> af[4];
#pragma unroll
            for (int o = 0; o < 4; ++o)
                af[o] = load_frag(reinterpret_cast<const T*>(patch + (pbase + (o >> 1) * PW + (o & 1)) * PATCH_PITCH + ks * 32) + h * 8);
#pragma unroll
            for (int k = 0; k < NTAP; ++k) {
                const int st_ = ks * NTAP + k;
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) mma(acc[tap_cls[k]][nt], af[tap_off[k]], bq[st_ % DEPTH][nt]);
                if (st_ + DEPTH < KS * NTAP) load_b(st_ + DEPTH, st_ % DEPTH);
                __builtin_amdgcn_sched_barrier(0);
            }
        }

        STAMP(4)
        if (chunk == NCH - 1) {
#pragma unroll
            for (int py = 0; py < 2; ++py) {
                if constexpr (EPI == EPI_BWD) {
#pragma unroll
                    for (int u = 0; u < OPL; ++u) {
                        int loff; (void)out_chunk_addr(cur, py, u, loff);
                        *reinterpret_cast<decltype(Vec16<T>::v)*>(mytile + loff) = prey[u];
                    }
                    if (py == 0) issue_y(cur, 1);
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                }
                auto epi_body = [&](auto checked) __attribute__((always_inline)) {
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) {
#pragma unroll
                        for (int px = 0; px < 2; ++px) {
#pragma unroll
                            for (int i = 0; i < 16; i += 2) {
                                const int row0 = acc_row(i, lane), row1 = acc_row(i + 1, lane);
                                T* c0 = reinterpret_cast<T*>(mytile + (2 * row0 + px) * OPITCH) + nt * 32 + r;
                                T* c1 = reinterpret_cast<T*>(mytile + (2 * row1 + px) * OPITCH) + nt * 32 + r;
                                constexpr bool CK_ = decltype(checked)::value;
                                const bool ok0 = !CK_ || (cur.b0 + ((wv0 * 32 + row0) >> (a.lth + a.ltw))) < a.B;
                                const bool ok1 = !CK_ || (cur.b0 + ((wv0 * 32 + row1) >> (a.lth + a.ltw))) < a.B;
                                epi_pair<T, EPI, CK_>(acc[py * 2 + px][nt][i], acc[py * 2 + px][nt][i + 1], c0, c1, ok0, ok1, esc[nt], esh[nt],
                                                      a.oslope, s1[nt], s2[nt]);
                                acc[py * 2 + px][nt][i] = ebv[nt]; acc[py * 2 + px][nt][i + 1] = ebv[nt];
                            }
                        }
                    }
                };
                if (EPI != EPI_FWD || cur.b0 + TB <= a.B) epi_body(std::false_type{}); else epi_body(std::true_type{});
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
                for (int u = 0; u < OPL; ++u) {
                    int loff; const int gi = out_chunk_addr(cur, py, u, loff);
                    const Vec16<T> v = *reinterpret_cast<const Vec16<T>*>(mytile + loff);
                    if (gi >= 0) *reinterpret_cast<Vec16<T>*>(at_bytes(a.out, (uint32_t)gi * (uint32_t)sizeof(T))) = v;
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // reads done before round 1 overwrites the tile
            }
        }
        STAMP(5)
        pi = npi; chunk = nchunk; cur = nxt; have = nhave;
    }
#ifdef VAE_PHASE_STAMPS
    if (a.dbg && lane == 0) { for (int k = 0; k < 6; ++k) a.dbg[(blockIdx.x * 4 + wave) * 6 + k] = tph[k]; }
#endif
#undef STAMP

    if constexpr (EPI != EPI_PLAIN) {
        __syncthreads();
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            float v1 = s1[nt].x + s1[nt].y, v2 = s2[nt].x + s2[nt].y;
            if constexpr (EPI == EPI_BWD) v2 = eis[nt] * v2 + exm[nt] * v1;   // sum dz*xhat from sum dz*y and sum dz
            v1 += __shfl_xor(v1, 32, 64); v2 += __shfl_xor(v2, 32, 64);
            if (h == 0) { red[((wave * NT + nt) * 32 + r) * 2] = v1; red[((wave * NT + nt) * 32 + r) * 2 + 1] = v2; }
        }
        __syncthreads();
        if (tid < NT * 32) {
            float v1 = 0.f, v2 = 0.f;
#pragma unroll
            for (int w = 0; w < 4; ++w) { v1 += red[((w * NT) * 32 + tid) * 2]; v2 += red[((w * NT) * 32 + tid) * 2 + 1]; }
            const int n0 = (vb % ntiles_n) * 32 * NT;
            double* st_ = a.stat + stat_rep() * 2 * Cout;
            unsafeAtomicAdd(&st_[n0 + tid], (double)v1);
            unsafeAtomicAdd(&st_[Cout + n0 + tid], (double)v2);
        }
    }
}
