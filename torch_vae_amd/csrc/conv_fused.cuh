// Fused backward of a ConvTranspose2d(k3,s2,p1,op1) layer whose high-resolution side has 32 channels (final_layer.0:
// 32->32, decoder.2: 64->32 of the reference, models.py:62-77): ONE pass over the layer's (dz, y) pair produces both
//   the input gradient   dz_prev[b,iy,ix,ci] = leaky'(z_prev) * sum_{ky,kx,co} g[b,2iy+ky-1,2ix+kx-1,co] * Wt[ci][co][ky][kx]
//   the weight gradient  dWt[ci][co][ky][kx]  = sum_{b,iy,ix} a_prev[b,iy,ix,ci] * g[b,2iy+ky-1,2ix+kx-1,co]
// where g = p0*dz + p1*y + p2 is the BatchNorm backward of this layer (applied while the patch is staged, never
// materialised) and a_prev = LeakyReLU(BN(y_prev)).  The separate kernels (down2_kernel + wgrad_kernel) each stream the
// (dz, y) pair - for final_layer.0 at the benchmark workload that is 2 x 536 MB of the step's 4.5 GB of kernel traffic;
// here it is read once.  These two layers are HBM-bound (32-channel tensors at the two largest resolutions); the deeper
// layers are MFMA-bound, their weight gradients are too large to live in registers, and they keep the separate kernels.
//
// Workgroup = 8 waves on an 8 x 16 low-res pixel tile of one image (patch 17 x 33 high-res pixels), persistent over tiles:
//   all waves   : next tile's raw (dz, y) chunks and y_prev rows prefetched in registers (5 + 5 + CLO/32 vectors per
//                 thread), transformed into the LDS patch after the barrier that retires the previous tile;
//                 weight gradient: the 9 taps (x low-res channel blocks) are split over the 8 waves, K = the tile's 128
//                 pixels read k-major from LDS (ds_read_b64_tr_b16), accumulators (2-3 tiles per wave) live in registers
//                 across all tiles of the workgroup and are written once, as one split-K slab per workgroup
//                 (reduce_slab_kernel sums them)
//   waves 0..3  : also the input gradient of 32 pixels each (18 MFMA per 32 output channels, weights from an LDS image
//                 loaded once per workgroup), epilogue (LeakyReLU', BatchNorm statistics of the previous layer) in place
//                 over the wave's own rows of the y_prev tile, 16-byte coalesced stores
// Splitting the weight-gradient tiles over all waves keeps every wave under 256 registers (two waves per SIMD).
#pragma once
#include "conv_mfma.cuh"
#include "conv_pipe.cuh"

template <typename T> struct ConvTFusedArgs {
    const T* dz; const T* y;            // this layer (high-res side) [B, 2Hs, 2Ws, 32]
    const float* gcoef;                 // rows p0,p1,p2 (stride 32) when fuse.mode == BNF_NONE
    BnFuse fuse;                        // BNF_BWD: derive p0..p2 from the batch statistics here (workgroup 0 records them)
    const T* wp;                        // packed dgrad weights [9][32/8][CLO][8]
    const T* yprev; const float* ocoef; // previous layer (low-res side) [B, Hs, Ws, CLO]; its block rows LC_* (stride CLO)
    T* dzprev; double* stat;            // outputs: dz of the previous layer, [sum dz | sum dz*xhat] (replicated, 2*CLO)
    float* slab;                        // [gridDim.x][9][CLO][32] partial weight gradients
    float slope;
    int B, Hs, Ws, n_tiles, tiles_x, tiles_y, rev;
    // RECOMP (final_layer.0 only): dz of this layer is NOT read - it is recomputed per patch pixel from the output conv's
    // logit gradient: dz[p][c] = leaky'(z[p][c]) * sum_t dlogit[p - off(t)] * wout[t][c]  (what convout_bwd_mfma_kernel
    // would have stored, rounded the same way), so the 32-channel gradient tensor at full resolution never exists in HBM.
    const float* dlogit; const float* gscale; float gmul;   // [B, 2Hs, 2Ws] f32; optional device scale; f16 gradient scale
    const float* wout;                  // output-conv weights, tap-major [9][32] f32
    const float* fcoef;                 // this layer's forward block (rows LC_SC / LC_SH, stride 32): z = sc*y + sh
};

// RECOMP = false: g = BN-backward(dz, y) from the stored pair.  RECOMP = true (CLO = 32): dz recomputed from dlogit.
template <typename T, int CLO, bool RECOMP = false>
__global__ __launch_bounds__(512, 2) void convt_bwd_fused_kernel(ConvTFusedArgs<T> a) {
    static_assert(sizeof(T) == 2, "16-bit storage only (f32 keeps the separate kernels)");
    static_assert(CLO == 32 || CLO == 64, "low-res channel count");
    typedef typename H16<T>::v8 T8;
    constexpr int TH = 8, TW = 16, NPX = TH * TW, PH = 2 * TH + 1, PW = 2 * TW + 1, NP = PH * PW;   // 128 px, 17 x 33 patch
    constexpr int NCHK = NP * 4, MAXI = (NCHK + 511) / 512;                                          // 2244 chunks, 5 per thread
    constexpr int GP = 80;                                   // patch pitch: 64 B of channels + 16 B pad
    constexpr int NB = CLO / 32;                             // 32-channel blocks of the low-res side
    constexpr int AP = CLO * 2 + 16;                         // y_prev / dz_prev tile pitch
    constexpr int ACH = CLO * 2 / 16, NYC = NPX * ACH / 512; // 16-byte chunks per low-res pixel; y_prev chunks per thread
    constexpr int NWT = (9 * NB + 7) / 8;                    // weight-gradient accumulator tiles per wave (tile idx = wave + 8j)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* gpatch = smem;                                     // [NP][GP]   g = BN-backward(dz, y), storage type
    char* atile = gpatch + NP * GP;                          // [NPX][AP]  a_prev = LeakyReLU(BN(y_prev)): the weight gradient's A operand
    char* ytile = atile + NPX * AP;                          // [NPX][AP]  raw y_prev, overwritten in place by dz_prev (rows of a dgrad wave are private to it)
    char* wlds = ytile + NPX * AP;                           // [9][4][CLO][8] dgrad weights
    float* cf = reinterpret_cast<float*>(wlds + 9 * 4 * CLO * 16);   // [3][32] p0,p1,p2 of this layer
    float* cfp = cf + 96;                                    // [2][CLO] scale, shift of the previous layer's BatchNorm
    float* red = cfp + 2 * CLO;                              // [4 waves][CLO][2]
    int2* gtab = reinterpret_cast<int2*>(red + 4 * CLO * 2); // [MAXI*512] tile-independent chunk geometry (kept out of the registers)
    char* dummy = reinterpret_cast<char*>(gtab + MAXI * 512); // 16 bytes: where the (masked) chunks beyond the patch are stored
    // RECOMP only: raw y patch (the chunks are staged untouched; the BatchNorm backward is applied per accumulator element
    // after the dz product) and the (PH+2) x (PW+2) patch of dlogit
    constexpr int DH = PH + 2, DW = PW + 2, NDL = DH * DW, NDLT = (NDL + 511) / 512;
    char* rawy = dummy + 16;                                 // [NP + pad to 18*32][GP]
    float* dlp = reinterpret_cast<float*>(rawy + 18 * 32 * GP);   // [DH][DW]
    float* cf7 = dlp + ((NDL + 3) & ~3);                     // [2][32] sc, sh of this layer's forward BatchNorm
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), r = lane & 31, h = lane >> 5;
    const int Hg = 2 * a.Hs, Wg = 2 * a.Ws;

    // ---- prologue: coefficients, weights, tile-independent chunk geometry
    if (tid < 32) {
        if (a.fuse.mode == BNF_BWD) bn_fused_channel(a.fuse, tid, blockIdx.x == 0, cf[tid], cf[32 + tid], cf[64 + tid]);
        else { cf[tid] = a.gcoef[tid]; cf[32 + tid] = a.gcoef[32 + tid]; cf[64 + tid] = a.gcoef[64 + tid]; }
    }
    if (tid >= 64 && tid < 64 + CLO) { const int n = tid - 64; cfp[n] = a.ocoef[LC_SC * CLO + n]; cfp[CLO + n] = a.ocoef[LC_SH * CLO + n]; }
    if constexpr (RECOMP) {
        if (tid >= 128 && tid < 160) { const int n = tid - 128; cf7[n] = a.fcoef[LC_SC * 32 + n]; cf7[32 + n] = a.fcoef[LC_SH * 32 + n]; }
        for (int i = tid; i < (18 * 32 - NP) * GP / 16; i += 512) *reinterpret_cast<f32x4*>(rawy + NP * GP + i * 16) = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    for (int i = tid; i < 9 * 4 * CLO; i += 512)
        *reinterpret_cast<T8*>(wlds + i * 16) = *reinterpret_cast<const T8*>(reinterpret_cast<const char*>(a.wp) + (size_t)i * 16);
    // chunk u of a thread: id = tid + 512u -> patch pixel id>>2, channel quarter id&3 = tid&3 (the same for every u).
    // Table entry: {element offset relative to the patch origin, LDS offset | top<<20 | left<<21 | beyond<<22}
#pragma unroll
    for (int u = 0; u < MAXI; ++u) {
        const int id = tid + 512 * u, pix = id >> 2, py = pix / PW, px = pix - py * PW;
        gtab[id] = make_int2((py * Wg + px) * 32 + (id & 3) * 8,
                             (id < NCHK ? pix * GP + (id & 3) * 16 : (int)(dummy - gpatch)) | ((py == 0) << 20) | ((px == 0) << 21) | ((id >= NCHK) << 22));
    }
    __syncthreads();

    auto tile_origin = [&](int t_, int& b, int& y0, int& x0) __attribute__((always_inline)) {
        const int t = a.rev ? a.n_tiles - 1 - t_ : t_;   // reversed walk: start with what the producer wrote last
        const int tx = t % a.tiles_x, ty = (t / a.tiles_x) % a.tiles_y;
        b = t / (a.tiles_x * a.tiles_y); y0 = ty * TH; x0 = tx * TW;
    };
    // prefetch registers (raw vectors)
    T8 pz[RECOMP ? 1 : MAXI], py_[MAXI], pyp[NYC];
    float pdl[RECOMP ? NDLT : 1];
    const float gs = RECOMP ? (a.gscale ? a.gscale[0] : 1.f) * a.gmul : 1.f;
    int pok = 0;   // validity bits of the prefetched chunks
    auto issue_dl = [&](int b, int y0, int x0, bool have) __attribute__((always_inline)) {
#pragma unroll
        for (int u = 0; u < (RECOMP ? NDLT : 0); ++u) {
            const int i = tid + 512 * u, rr = i / DW, cc = i - rr * DW, gy = 2 * y0 - 2 + rr, gx = 2 * x0 - 2 + cc;
            const bool in = have && i < NDL && gy >= 0 && gy < Hg && gx >= 0 && gx < Wg;
            const float v = a.dlogit[in ? ((size_t)b * Hg + gy) * Wg + gx : 0];
            pdl[u] = in ? v * gs : 0.f;
        }
    };
    auto issue = [&](int t) __attribute__((always_inline)) {
        int b, y0, x0; tile_origin(t, b, y0, x0);
        const int base = ((b * Hg + 2 * y0 - 1) * Wg + 2 * x0 - 1) * 32;
        const int tmask = (y0 == 0 ? 1 << 20 : 0) | (x0 == 0 ? 1 << 21 : 0) | (1 << 22);
        pok = 0;
#pragma unroll
        for (int u = 0; u < MAXI; ++u) {
            const int2 e = gtab[tid + 512 * u];
            const bool ok = (e.y & tmask) == 0;
            pok |= ok ? (1 << u) : 0;
            const uint32_t off = ok ? (uint32_t)(base + e.x) * 2u : 0u;
            if constexpr (!RECOMP) pz[u] = *reinterpret_cast<const T8*>(at_bytes(a.dz, off));
            py_[u] = *reinterpret_cast<const T8*>(at_bytes(a.y, off));
        }
        issue_dl(b, y0, x0, true);
#pragma unroll
        for (int u = 0; u < NYC; ++u) {
            const int id = tid + 512 * u, R = id / ACH, qq = id - R * ACH;
            const uint32_t off = (uint32_t)(((b * a.Hs + y0 + (R >> 4)) * a.Ws + x0 + (R & 15)) * CLO + qq * 8) * 2u;
            pyp[u] = *reinterpret_cast<const T8*>(at_bytes(a.yprev, off));
        }
    };

    // ---- per-role state (all waves: weight-gradient tiles wave + 8j; waves 0..3 also the input gradient of 32 pixel rows)
    const int wq = wave & 3;
    const int g4 = lane >> 4, li = lane & 15, q = li >> 2, p = li & 3;          // lane geometry of the transposed LDS reads
    const int Rr = wq * 32 + r, pbase = (2 * (Rr >> 4)) * PW + 2 * (Rr & 15);   // dgrad: this lane's pixel row -> patch pixel of tap (0,0)
    float esc[NB], esh[NB], eis[NB], exm[NB];   // previous layer's forward coefficients of channel nb*32 + r
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        const int n = nb * 32 + r;
        esc[nb] = a.ocoef[LC_SC * CLO + n]; esh[nb] = a.ocoef[LC_SH * CLO + n];
        eis[nb] = a.ocoef[LC_INVSTD * CLO + n]; exm[nb] = a.ocoef[LC_XM * CLO + n];
    }
    // weight-gradient tiles of this wave: idx = wave + 8j < 9*NB -> tap idx / NB, low-res channel block idx % NB (= wave % NB)
    const bool cib1 = NB == 2 && (wave & 1);
    const int acol = ((cib1 ? 32 : 0) + 16 * (g4 & 1) + 4 * p) * 2, bcol = (16 * (g4 & 1) + 4 * p) * 2;
    f32x16 wacc[NWT];
#pragma unroll
    for (int j = 0; j < NWT; ++j)
#pragma unroll
        for (int i = 0; i < 16; ++i) wacc[j][i] = 0.f;
    f32x2 s1[NB], s2[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) { s1[nb] = f32x2{0.f, 0.f}; s2[nb] = f32x2{0.f, 0.f}; }

    // g = p0*dz + p1*y + p2 on packed pairs; the thread's 8 channels of p0,p1,p2 are re-read from LDS per tile, so
    // that they are not live across the matrix phase
    auto xform = [&](const T8& vz, const T8& vy, const f32x2* k0, const f32x2* k1, const f32x2* k2) __attribute__((always_inline)) {
        T8 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const f32x2 x0 = {(float)vz[2 * e], (float)vz[2 * e + 1]}, x1 = {(float)vy[2 * e], (float)vy[2 * e + 1]};
            const f32x2 z = x0 * k0[e] + (x1 * k1[e] + k2[e]);
            o[2 * e] = (T)z.x; o[2 * e + 1] = (T)z.y;
        }
        return o;
    };

    int t = blockIdx.x;
    if (t < a.n_tiles) issue(t);
    for (; t < a.n_tiles; t += gridDim.x) {
        int b, y0, x0; tile_origin(t, b, y0, x0);
        __syncthreads();                                   // (A) previous tile fully consumed
        // ---- stage this tile; as each register pair becomes free, request the same chunk of the next tile
        const int tn = t + (int)gridDim.x;
        const bool nh = tn < a.n_tiles;
        int nb_ = b, ny0 = y0, nx0 = x0;
        if (nh) tile_origin(tn, nb_, ny0, nx0);
        const int nbase = ((nb_ * Hg + 2 * ny0 - 1) * Wg + 2 * nx0 - 1) * 32;
        const int ntmask = (ny0 == 0 ? 1 << 20 : 0) | (nx0 == 0 ? 1 << 21 : 0) | (1 << 22);
        int npok = 0;
        f32x2 k0[4], k1[4], k2[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int c = (tid & 3) * 8 + 2 * e;
            k0[e] = *reinterpret_cast<const f32x2*>(cf + c); k1[e] = *reinterpret_cast<const f32x2*>(cf + 32 + c); k2[e] = *reinterpret_cast<const f32x2*>(cf + 64 + c);
        }
        const int cur_pok = pok;
#pragma unroll
        for (int u = 0; u < MAXI; ++u) {
            const int2 e = gtab[tid + 512 * u];
            if constexpr (RECOMP) {
                // raw y (zeros outside the image): rawy shares the patch geometry; chunks beyond the patch go to the dummy slot
                T8 o = py_[u];
                if (!((pok >> u) & 1)) o = T8{0, 0, 0, 0, 0, 0, 0, 0};
                *reinterpret_cast<T8*>(((e.y >> 22) & 1 ? gpatch : rawy) + (e.y & 0xfffff)) = o;
            } else {
                T8 o = xform(pz[u], py_[u], k0, k1, k2);
                if (!((pok >> u) & 1)) o = T8{0, 0, 0, 0, 0, 0, 0, 0};
                *reinterpret_cast<T8*>(gpatch + (e.y & 0xfffff)) = o;   // (chunks beyond the patch land in the dummy slot)
            }
            const bool ok = nh & ((e.y & ntmask) == 0);
            npok |= ok ? (1 << u) : 0;
            const uint32_t off = ok ? (uint32_t)(nbase + e.x) * 2u : 0u;
            if constexpr (!RECOMP) pz[u] = *reinterpret_cast<const T8*>(at_bytes(a.dz, off));
            py_[u] = *reinterpret_cast<const T8*>(at_bytes(a.y, off));
        }
        pok = npok;
        if constexpr (RECOMP) {
#pragma unroll
            for (int u = 0; u < NDLT; ++u) { const int i = tid + 512 * u; if (i < NDL) dlp[i] = pdl[u]; }
            issue_dl(nb_, ny0, nx0, nh);
        }
#pragma unroll
        for (int u = 0; u < NYC; ++u) {
            const int id = tid + 512 * u, R = id / ACH, qq = id - R * ACH;
            *reinterpret_cast<T8*>(ytile + R * AP + qq * 16) = pyp[u];
            T8 av;   // a_prev for the weight gradient (the chunk's 8 channels: qq*8 ..)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const f32x2 sc2 = *reinterpret_cast<const f32x2*>(cfp + qq * 8 + 2 * e), sh2 = *reinterpret_cast<const f32x2*>(cfp + CLO + qq * 8 + 2 * e);
                f32x2 z = f32x2{(float)pyp[u][2 * e], (float)pyp[u][2 * e + 1]} * sc2 + sh2;
                const f32x2 zs = z * a.slope;
                z.x = fmaxf(z.x, zs.x); z.y = fmaxf(z.y, zs.y);
                av[2 * e] = (T)z.x; av[2 * e + 1] = (T)z.y;
            }
            *reinterpret_cast<T8*>(atile + R * AP + qq * 16) = av;
            const uint32_t off = nh ? (uint32_t)(((nb_ * a.Hs + ny0 + (R >> 4)) * a.Ws + nx0 + (R & 15)) * CLO + qq * 8) * 2u : 0u;
            pyp[u] = *reinterpret_cast<const T8*>(at_bytes(a.yprev, off));
        }
        __syncthreads();                                   // (B) patch (RECOMP: raw y + dlogit) and y_prev tile published
        if constexpr (RECOMP) {
            // ---- dz of this layer for the 17 x 33 patch pixels: 18 blocks of 32 pixels over the 8 waves, ONE k-step each
            // (K = 9 taps, padded to 16), the same product convout_bwd_mfma_kernel forms; then LeakyReLU', the storage
            // rounding the stored tensor would have had, and the BatchNorm backward, per accumulator element
            Frag<T> wfrag;   // B[k = tap][n = channel r]
#pragma unroll
            for (int j = 0; j < 8; ++j) { const int tp = 8 * h + j; wfrag.v[j] = (T)(tp < 9 ? a.wout[tp * 32 + r] : 0.f); }
            const float sc7 = cf7[r], sh7 = cf7[32 + r], q0 = cf[r], q1 = cf[32 + r], q2 = cf[64 + r];
            const bool top_out = y0 == 0, left_out = x0 == 0;
#pragma unroll 1
            for (int m = wave; m < 18; m += 8) {
                const int pm = m * 32 + r, pmc = pm < NP ? pm : NP - 1, ppy = pmc / PW, ppx = pmc - ppy * PW;
                Frag<T> af;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int tp = 8 * h + j, tt = tp < 9 ? tp : 0;
                    const float v = dlp[(ppy + 2 - tt / 3) * DW + (ppx + 2 - tt % 3)];
                    af.v[j] = (T)(tp < 9 ? v : 0.f);
                }
                f32x16 da;
#pragma unroll
                for (int i = 0; i < 16; ++i) da[i] = 0.f;
                mma(da, af, wfrag);
                // accumulator rows i, i+1 are consecutive patch pixels: processed as a pair so that the storage conversion is
                // the packed instruction the staging transform of the stored-dz path uses (identical roundings, bit-identical g)
#pragma unroll
                for (int i = 0; i < 16; i += 2) {
                    const int pe = m * 32 + acc_row(i, lane);     // (acc_row(i + 1) = acc_row(i) + 1; NP is odd, pe is even)
                    if (pe < NP) {
                        const bool two = pe + 1 < NP;
                        const int ey0 = pe / PW, ex0 = pe - ey0 * PW, ey1 = (pe + 1) / PW, ex1 = pe + 1 - ey1 * PW;
                        const f32x2 yv = {tofloat(*reinterpret_cast<const T*>(rawy + pe * GP + r * 2)), tofloat(*reinterpret_cast<const T*>(rawy + (pe + 1) * GP + r * 2))};
                        const f32x2 z = yv * sc7 + sh7;
                        f32x2 dzv = {tofloat(fromfloat<T>(z.x > 0.f ? da[i] : da[i] * a.slope)), tofloat(fromfloat<T>(z.y > 0.f ? da[i + 1] : da[i + 1] * a.slope))};
                        f32x2 g = dzv * q0 + (yv * q1 + q2);
                        if ((top_out && ey0 == 0) || (left_out && ex0 == 0)) g.x = 0.f;     // patch pixels outside the image carry no gradient
                        if ((top_out && ey1 == 0) || (left_out && ex1 == 0)) g.y = 0.f;
                        T o0, o1;
                        (void)round_pair<T>(g, o0, o1);
                        *reinterpret_cast<T*>(gpatch + pe * GP + r * 2) = o0;
                        if (two) *reinterpret_cast<T*>(gpatch + (pe + 1) * GP + r * 2) = o1;
                    }
                }
            }
            (void)cur_pok;
            __syncthreads();                               // (C) g patch published
        }

        if (wave < 4) {
            // ---- input gradient of pixel rows wq*32 .. wq*32+31
            f32x16 dacc[NB];
#pragma unroll
            for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                for (int i = 0; i < 16; ++i) dacc[nb][i] = 0.f;
#pragma unroll
            for (int tp = 0; tp < 9; ++tp) {
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    const Frag<T> af = load_frag(reinterpret_cast<const T*>(gpatch + (pbase + (tp / 3) * PW + (tp % 3)) * GP + ks * 32) + h * 8);
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb) {
                        const Frag<T> bf = load_frag(reinterpret_cast<const T*>(wlds + (((tp * 4 + 2 * ks + h) * CLO + nb * 32 + r) * 16)));
                        mma(dacc[nb], af, bf);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);   // keep the fragment loads of later taps from being hoisted (register budget)
            }
            // epilogue: dz_prev = leaky'(z_prev) * acc, statistics of the stored values; rows of this wave only
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
#pragma unroll
                for (int i = 0; i < 16; i += 2) {
                    const int R0 = wq * 32 + acc_row(i, lane), R1 = wq * 32 + acc_row(i + 1, lane), n = nb * 32 + r;
                    T* c0 = reinterpret_cast<T*>(ytile + R0 * AP + n * 2); T* c1 = reinterpret_cast<T*>(ytile + R1 * AP + n * 2);
                    const f32x2 yv = {tofloat(*c0), tofloat(*c1)};
                    const f32x2 z = yv * esc[nb] + esh[nb];
                    f32x2 g = {dacc[nb][i], dacc[nb][i + 1]};
                    g.x = z.x > 0.f ? g.x : g.x * a.slope; g.y = z.y > 0.f ? g.y : g.y * a.slope;
                    const f32x2 dzv = round_pair<T>(g, *c0, *c1);
                    s1[nb] += dzv; s2[nb] += dzv * yv;
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // own LDS writes landed (wave-private rows)
#pragma unroll
            for (int u = 0; u < ACH / 2; ++u) {
                const int id = lane + 64 * u, row = id / ACH, qq = id - row * ACH, R = wq * 32 + row;
                const T8 v = *reinterpret_cast<const T8*>(ytile + R * AP + qq * 16);
                const uint32_t off = (uint32_t)(((b * a.Hs + y0 + (R >> 4)) * a.Ws + x0 + (R & 15)) * CLO + qq * 8) * 2u;
                *reinterpret_cast<T8*>(at_bytes(a.dzprev, off)) = v;
            }
        }
        {
            // ---- weight gradient: K = the tile's 128 pixels, A = a_prev^T (k-major reads of the a_prev tile), B = g at the tap's
            // patch pixels (k-major reads of the patch).  Fragments are re-read per tile: LDS has the bandwidth, registers do not.
#pragma unroll
            for (int j = 0; j < NWT; ++j) {
                const int idx = wave + 8 * j;
                if (idx < 9 * NB) {   // wave-uniform
                    const int tp = idx / NB, toff = (tp / 3) * PW + (tp % 3);
#pragma unroll
                    for (int ks = 0; ks < 8; ++ks) {
                        const int kk0 = ks * 16 + 8 * (g4 >> 1) + q, kk1 = kk0 + 4;
                        const int gb0 = (2 * (kk0 >> 4)) * PW + 2 * (kk0 & 15), gb1 = (2 * (kk1 >> 4)) * PW + 2 * (kk1 & 15);
                        const Frag<T> af = frag_tr16<T>(atile + kk0 * AP + acol, atile + kk1 * AP + acol);
                        const Frag<T> bf = frag_tr16<T>(gpatch + (gb0 + toff) * GP + bcol, gpatch + (gb1 + toff) * GP + bcol);
                        mma(wacc[j], af, bf);
                        if (ks & 1) __builtin_amdgcn_sched_barrier(0);   // at most two k-steps of fragments in flight
                    }
                }
            }
        }
    }

    // ---- workgroup results: BatchNorm statistics of the previous layer (dgrad waves), weight-gradient slab (wgrad waves)
    if (wave < 4) {
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            float v1 = s1[nb].x + s1[nb].y, v2 = s2[nb].x + s2[nb].y;
            v2 = eis[nb] * v2 + exm[nb] * v1;                    // sum dz*xhat from sum dz*y and sum dz
            v1 += __shfl_xor(v1, 32, 64); v2 += __shfl_xor(v2, 32, 64);
            if (h == 0) { red[((wq * NB + nb) * 32 + r) * 2] = v1; red[((wq * NB + nb) * 32 + r) * 2 + 1] = v2; }
        }
    }
    {
#pragma unroll
        for (int j = 0; j < NWT; ++j) {
            const int idx = wave + 8 * j;
            if (idx < 9 * NB) {
                const int tp = idx / NB, cib = idx % NB;
#pragma unroll
                for (int i = 0; i < 16; ++i)
                    a.slab[(((size_t)blockIdx.x * 9 + tp) * CLO + cib * 32 + acc_row(i, lane)) * 32 + r] = wacc[j][i];
            }
        }
    }
    __syncthreads();
    if (tid < CLO) {
        const int nb = tid >> 5, c = tid & 31;
        float v1 = 0.f, v2 = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) { v1 += red[((w * NB + nb) * 32 + c) * 2]; v2 += red[((w * NB + nb) * 32 + c) * 2 + 1]; }
        double* st_ = a.stat + stat_rep() * 2 * CLO;
        unsafeAtomicAdd(&st_[tid], (double)v1);
        unsafeAtomicAdd(&st_[CLO + tid], (double)v2);
    }
}

// LDS bytes of convt_bwd_fused_kernel<T, CLO, RECOMP>
static inline size_t convt_fused_lds(int CLO, bool recomp = false) {
    return (recomp ? (size_t)18 * 32 * 80 + (size_t)((19 * 35 + 3) & ~3) * 4 + 64 * 4 : 0) + (size_t)(17 * 33) * 80 + 2 * (size_t)128 * (CLO * 2 + 16) + (size_t)9 * 4 * CLO * 16 + 96 * 4 + (size_t)2 * CLO * 4 + (size_t)4 * CLO * 2 * 4 + (size_t)5 * 512 * 8 + 16;
}

// ---------------------------------------------------------------------------------------------------------------------
// Fused backward of a Conv2d(k3,s2,p1) layer with a 32-channel HIGH-res input side and 64 output channels (encoder.1 of the
// reference: Conv2d(32, 64), models.py:41-51): one pass over the layer's (dz, y) pair (low-res) and the previous layer's y
// (high-res) produces
//   the input gradient   dz_prev[b, 2i+py, 2j+px, ci] = leaky'(z_prev) * sum_{taps of parity (py,px)} g[b, i+di, j+dj, co] * W[co][ci][t]
//   the weight gradient  dW[co][ci][ky][kx] = sum_{b,oy,ox} g[b,oy,ox,co] * a_prev[b, 2oy+ky-1, 2ox+kx-1, ci]
// with g = p0*dz + p1*y + p2 (BatchNorm backward of this layer) and a_prev = LeakyReLU(BN(y_prev)).  The separate kernels
// (up2_kernel + wgrad_kernel) read y_prev and the (dz, y) pair twice; here once (+ the one-pixel halo of y_prev).
//
// Workgroup = 8 waves on an 8 x 8 low-res tile of one image = a 16 x 16 high-res output tile, persistent over tiles:
//   all waves  : next tile's raw chunks prefetched in registers; staged after the barrier that retires the previous tile:
//                g (9 x 9 low-res patch: the "up" taps reach one pixel right / down), a_prev (17 x 17 high-res patch with the
//                top / left halo the weight gradient's taps need) and the raw y_prev rows of the output tile
//   waves 0..3 : input gradient of 32 low-res pixels each (wave & 1) for one group of output parities (wave >> 1: the (odd,odd)
//                class with its 4 taps, or the other three classes with 5 taps) - disjoint outputs, no cross-wave sum; the f32
//                accumulators go to an LDS tile [pixel][channel]
//   waves 2..7 : weight gradient, the 18 (tap, 32-channel block of co) tiles split 3 per wave (waves 2, 3 carry the light
//                parity group of the input gradient as well), K = the tile's 64 low-res pixels read k-major
//                (ds_read_b64_tr_b16); accumulators persist over the workgroup's tiles -> one slab per workgroup
//   all waves  : after a barrier, the epilogue in CHUNK layout (a thread owns 8 channels of a pixel: packed math, every thread
//                busy, instead of the accumulator layout's per-element 2-byte LDS cells on the four dgrad waves): LeakyReLU',
//                storage rounding, BatchNorm statistics of the previous layer, 16-byte stores straight to HBM as 1-KiB rows
template <typename T> struct ConvFusedArgs {
    const T* dz; const T* y;            // this layer (low-res side) [B, Hs, Ws, 64]
    const float* gcoef; BnFuse fuse;    // p0,p1,p2 rows (stride 64) / derive them here (BNF_BWD; workgroup 0 records them)
    const T* wp;                        // packed dgrad weights [9][64/8][32][8]
    const T* yprev; const float* ocoef; // previous layer (high-res side) [B, 2Hs, 2Ws, 32]; its block rows LC_* (stride 32)
    T* dzprev; double* stat;            // outputs: dz of the previous layer; [sum dz | sum dz*xhat] (replicated, 2*32)
    float* slab;                        // [gridDim.x][9][64][32] partial weight gradients
    float slope;
    int B, Hs, Ws, n_tiles, tiles_x, tiles_y, rev;
    int ablate;                         // diagnostics (timing only, results wrong): 1 no matrix phase, 2 no epilogue, 4 no staging, 8 no reloads
};

template <typename T>
__global__ __launch_bounds__(512, 2) void conv_bwd_fused_kernel(ConvFusedArgs<T> a) {
    static_assert(sizeof(T) == 2, "16-bit storage only (f32 keeps the separate kernels)");
    typedef typename H16<T>::v8 T8;
    constexpr int CLO = 64, TH = 8, TW = 8, NLO = TH * TW;                 // 64 low-res pixels
    constexpr int LPH = TH + 1, LPW = TW + 1, NLP = LPH * LPW;             // 9 x 9 low-res patch of g
    constexpr int HT = 2 * TH, WT = 2 * TW, NHI = HT * WT;                 // 16 x 16 high-res output tile
    constexpr int HPH = HT + 1, HPW = WT + 1, NHP = HPH * HPW;             // 17 x 17 high-res patch of a_prev
    constexpr int LP = CLO * 2 + 16, HP = 80;                              // LDS pitches (bytes per pixel)
    constexpr int NGC = NLP * 8, NG = (NGC + 511) / 512;                   // g chunks (8 per pixel): 648, 2 per thread
    constexpr int NAC = NHP * 4, NA = (NAC + 511) / 512;                   // a_prev chunks (4 per pixel): 1156, 3 per thread
    constexpr int NWT = 3;                                                 // weight-gradient tiles per wgrad wave (18 over waves 2..7)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* gpatch = smem;                                  // [NLP][LP]  g
    char* apatch = gpatch + NLP * LP;                     // [NHP][HP]  a_prev (zero outside the image)
    char* ytile = apatch + NHP * HP;                      // [NHI][HP]  raw y_prev of the output tile -> dz_prev in place
    char* wlds = ytile + NHI * HP;                        // [9][8][32][8] dgrad weights
    float* cfg = reinterpret_cast<float*>(wlds + 9 * 8 * 32 * 16);   // [3][64] p0,p1,p2 of this layer
    float* cfa = cfg + 3 * CLO;                           // [2][32] scale, shift of the previous layer's BatchNorm
    float* red = cfa + 64;                                // [4 waves][32][2]
    // tile-independent chunk geometry, kept out of the registers: {element offset relative to the patch origin,
    // LDS offset | py << 16 | px << 24}; g chunks first (NG * 512), then a_prev chunks (NA * 512)
    int2* ctab = reinterpret_cast<int2*>(red + 4 * 32 * 2);
    constexpr int FP = 32 * 4 + 16;                       // f32 accumulator tile pitch
    char* ftile = reinterpret_cast<char*>(ctab + (NG + NA) * 512);   // [NHI][FP] input-gradient accumulators (f32)
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), r = lane & 31, h = lane >> 5;
    const int Hg = 2 * a.Hs, Wg = 2 * a.Ws;

    // ---- prologue
#pragma unroll
    for (int u = 0; u < NG; ++u) {
        const int id = tid + 512 * u, pix = id >> 3, py = pix / LPW, px = pix - py * LPW;
        ctab[id] = make_int2((py * a.Ws + px) * CLO + (id & 7) * 8, (id < NGC ? pix * LP + (id & 7) * 16 : 0xffff) | (py << 16) | (px << 24));
    }
#pragma unroll
    for (int u = 0; u < NA; ++u) {
        const int id = tid + 512 * u, pix = id >> 2, py = pix / HPW, px = pix - py * HPW;
        ctab[NG * 512 + id] = make_int2((py * Wg + px) * 32 + (id & 3) * 8, (id < NAC ? pix * HP + (id & 3) * 16 : 0xffff) | (py << 16) | (px << 24));
    }
    if (tid < CLO) {
        if (a.fuse.mode == BNF_BWD) bn_fused_channel(a.fuse, tid, blockIdx.x == 0, cfg[tid], cfg[CLO + tid], cfg[2 * CLO + tid]);
        else { cfg[tid] = a.gcoef[tid]; cfg[CLO + tid] = a.gcoef[CLO + tid]; cfg[2 * CLO + tid] = a.gcoef[2 * CLO + tid]; }
    }
    if (tid >= 64 && tid < 96) { const int n = tid - 64; cfa[n] = a.ocoef[LC_SC * 32 + n]; cfa[32 + n] = a.ocoef[LC_SH * 32 + n]; }
    for (int i = tid; i < 9 * 8 * 32; i += 512)
        *reinterpret_cast<T8*>(wlds + i * 16) = *reinterpret_cast<const T8*>(reinterpret_cast<const char*>(a.wp) + (size_t)i * 16);

    auto tile_origin = [&](int t_, int& b, int& y0, int& x0) __attribute__((always_inline)) {
        const int t = a.rev ? a.n_tiles - 1 - t_ : t_;
        const int tx = t % a.tiles_x, ty = (t / a.tiles_x) % a.tiles_y;
        b = t / (a.tiles_x * a.tiles_y); y0 = ty * TH; x0 = tx * TW;
    };
    // chunk geometry (tile-independent): g chunk id = tid + 512u -> low-res patch pixel id>>3, channel eighth id&7;
    // a chunk id -> high-res patch pixel id>>2, channel quarter id&3
    T8 pz[NG], pyy[NG], pa[NA];
    int gok = 0, aok = 0;   // validity bits of the prefetched chunks
    auto issue = [&](int t, bool have) __attribute__((always_inline)) {
        int b, y0, x0; tile_origin(t, b, y0, x0);
        gok = 0; aok = 0;
        const int gbase = ((b * a.Hs + y0) * a.Ws + x0) * CLO, abase = ((b * Hg + 2 * y0 - 1) * Wg + 2 * x0 - 1) * 32;
#pragma unroll
        for (int u = 0; u < NG; ++u) {
            const int2 e = ctab[tid + 512 * u];
            const int py = (e.y >> 16) & 0xff, px = (e.y >> 24) & 0xff;
            const bool ok = have && (e.y & 0xffff) != 0xffff && y0 + py < a.Hs && x0 + px < a.Ws;     // bottom / right halo beyond the image: zero
            gok |= ok ? (1 << u) : 0;
            const uint32_t off = ok ? (uint32_t)(gbase + e.x) * 2u : 0u;
            pz[u] = *reinterpret_cast<const T8*>(at_bytes(a.dz, off));
            pyy[u] = *reinterpret_cast<const T8*>(at_bytes(a.y, off));
        }
#pragma unroll
        for (int u = 0; u < NA; ++u) {
            const int2 e = ctab[NG * 512 + tid + 512 * u];
            const int py = (e.y >> 16) & 0xff, px = (e.y >> 24) & 0xff;
            const bool ok = have && (e.y & 0xffff) != 0xffff && (y0 > 0 || py > 0) && (x0 > 0 || px > 0);   // top / left halo outside the image: zero
            aok |= ok ? (1 << u) : 0;
            const uint32_t off = ok ? (uint32_t)(abase + e.x) * 2u : 0u;
            pa[u] = *reinterpret_cast<const T8*>(at_bytes(a.yprev, off));
        }
    };

    // ---- per-role state
    const int wq = wave & 3;
    const int g4 = lane >> 4, li = lane & 15, q = li >> 2, p = li & 3;
    // dgrad (waves 0..3): m tile wq & 1 (low-res pixel row R = m*32 + r), parity group wq >> 1
    const int Rr = (wq & 1) * 32 + r, pbase = (Rr >> 3) * LPW + (Rr & 7);
    f32x2 s1[4], s2[4];     // statistics of this thread's 8 channels ((tid & 3) * 8 ..): sum dz, sum dz*y
#pragma unroll
    for (int e = 0; e < 4; ++e) { s1[e] = f32x2{0.f, 0.f}; s2[e] = f32x2{0.f, 0.f}; }
    // wgrad (waves 2..7, wj = wave - 2): tiles idx = wj + 6j < 18 -> tap idx >> 1, co block idx & 1 (= wj & 1)
    const int wj = wave - 2;
    const int acol = ((wj & 1) * 32 + 16 * (g4 & 1) + 4 * p) * 2, bcol = (16 * (g4 & 1) + 4 * p) * 2;
    f32x16 wacc[NWT];
#pragma unroll
    for (int j = 0; j < NWT; ++j)
#pragma unroll
        for (int i = 0; i < 16; ++i) wacc[j][i] = 0.f;
    constexpr int tap_cls[9] = {0, 1, 1, 2, 2, 3, 3, 3, 3};     // out(2i+py,2j+px) <- in(i+di,j+dj) * W[tap]   (as up2_kernel)
    constexpr int tap_t[9] = {4, 5, 3, 7, 1, 8, 6, 2, 0};
    constexpr int tap_off[9] = {0, 0, 1, 0, 2, 0, 1, 2, 3};     // di*2 + dj

    __syncthreads();
    int t = blockIdx.x;
    if (t < a.n_tiles) issue(t, true);
    for (; t < a.n_tiles; t += gridDim.x) {
        int b, y0, x0; tile_origin(t, b, y0, x0);
        __syncthreads();                                   // (A) previous tile fully consumed
        // ---- stage: g patch (BN backward on packed pairs), a_prev patch (BN + LeakyReLU), raw y_prev rows of the output tile
        if (!VAE_ABLATE(a.ablate, 4)) {
            const int cg0 = (tid & 7) * 8;
#pragma unroll
            for (int u = 0; u < NG; ++u) {
                const int lo_ = ctab[tid + 512 * u].y & 0xffff;
                T8 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int c = cg0 + 2 * e;
                    const f32x2 k0 = *reinterpret_cast<const f32x2*>(cfg + c), k1 = *reinterpret_cast<const f32x2*>(cfg + CLO + c), k2 = *reinterpret_cast<const f32x2*>(cfg + 2 * CLO + c);
                    const f32x2 x0_ = {(float)pz[u][2 * e], (float)pz[u][2 * e + 1]}, x1_ = {(float)pyy[u][2 * e], (float)pyy[u][2 * e + 1]};
                    const f32x2 z = x0_ * k0 + (x1_ * k1 + k2);
                    o[2 * e] = (T)z.x; o[2 * e + 1] = (T)z.y;
                }
                if (!((gok >> u) & 1)) o = T8{0, 0, 0, 0, 0, 0, 0, 0};
                if (lo_ != 0xffff) *reinterpret_cast<T8*>(gpatch + lo_) = o;
            }
            const int ca0 = (tid & 3) * 8;
#pragma unroll
            for (int u = 0; u < NA; ++u) {
                const int ey_ = ctab[NG * 512 + tid + 512 * u].y, lo_ = ey_ & 0xffff, py = (ey_ >> 16) & 0xff, px = (ey_ >> 24) & 0xff;
                T8 av;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const f32x2 sc2 = *reinterpret_cast<const f32x2*>(cfa + ca0 + 2 * e), sh2 = *reinterpret_cast<const f32x2*>(cfa + 32 + ca0 + 2 * e);
                    f32x2 z = f32x2{(float)pa[u][2 * e], (float)pa[u][2 * e + 1]} * sc2 + sh2;
                    const f32x2 zs = z * a.slope;
                    z.x = fmaxf(z.x, zs.x); z.y = fmaxf(z.y, zs.y);
                    av[2 * e] = (T)z.x; av[2 * e + 1] = (T)z.y;
                }
                if (!((aok >> u) & 1)) av = T8{0, 0, 0, 0, 0, 0, 0, 0};
                if (lo_ != 0xffff) {
                    *reinterpret_cast<T8*>(apatch + lo_) = av;
                    if (py >= 1 && px >= 1) *reinterpret_cast<T8*>(ytile + lo_ - (HPW + py) * HP) = pa[u];   // raw y_prev of the tile: pixel (py-1)*WT + px-1
                }
            }
        }
        const int tn = t + (int)gridDim.x;
        if (!VAE_ABLATE(a.ablate, 8)) issue(tn < a.n_tiles ? tn : t, tn < a.n_tiles);    // next tile's loads fly during the matrix phase
        __syncthreads();                                   // (B) patches published

        if (wave < 4 && !VAE_ABLATE(a.ablate, 1)) {
            // ---- input gradient: 32 low-res pixels x the output parities of this wave's group
            const bool grpB = wq < 2;                       // group B (waves 0,1): classes 0,1,2 (taps 0..4); group A (waves 2,3): class 3 (taps 5..8)
            f32x16 dacc[3];
#pragma unroll
            for (int cI = 0; cI < 3; ++cI)
#pragma unroll
                for (int i = 0; i < 16; ++i) dacc[cI][i] = 0.f;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                Frag<T> af[4];
#pragma unroll
                for (int o = 0; o < 4; ++o)
                    af[o] = load_frag(reinterpret_cast<const T*>(gpatch + (pbase + (o >> 1) * LPW + (o & 1)) * LP + ks * 32) + h * 8);
                if (grpB) {
#pragma unroll
                    for (int k = 0; k < 5; ++k) {
                        const Frag<T> bf = load_frag(reinterpret_cast<const T*>(wlds + (((tap_t[k] * 8 + 2 * ks + h) * 32 + r) * 16)));
                        mma(dacc[tap_cls[k]], af[tap_off[k]], bf);
                    }
                } else {
#pragma unroll
                    for (int k = 5; k < 9; ++k) {
                        const Frag<T> bf = load_frag(reinterpret_cast<const T*>(wlds + (((tap_t[k] * 8 + 2 * ks + h) * 32 + r) * 16)));
                        mma(dacc[0], af[tap_off[k]], bf);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            // epilogue in place: class c = py*2 + px -> high-res pixel (2*ty + py, 2*tx + px) of the tile
            // accumulator element i of lane (r, h) is low-res pixel (4m + (i >> 2), 4h + (i & 3)) of the tile: class (py, px) puts it
            // at high-res pixel (2*row + py, 2*col + px): a per-lane base plus a compile-time constant
            char* fbase = ftile + (((wq & 1) * 8) * WT + 8 * h) * FP + r * 4;
            auto put_class = [&](const f32x16& acc, int cls) __attribute__((always_inline)) {
                const int cpy = cls >> 1, cpx = cls & 1;
#pragma unroll
                for (int i = 0; i < 16; ++i)
                    *reinterpret_cast<float*>(fbase + ((2 * (i >> 2) + cpy) * WT + 2 * (i & 3) + cpx) * FP) = acc[i];
            };
            if (grpB) { put_class(dacc[0], 0); put_class(dacc[1], 1); put_class(dacc[2], 2); }
            else put_class(dacc[0], 3);
        }
        if (wave >= 2 && !VAE_ABLATE(a.ablate, 1)) {
            // ---- weight gradient: A = g^T (k-major reads of the patch's 64 centre pixels), B = a_prev at the tap's high-res pixels
#pragma unroll
            for (int j = 0; j < NWT; ++j) {
                const int idx = wj + 6 * j;
                {
                    const int tp = idx >> 1, toff = (tp / 3) * HPW + (tp % 3);
#pragma unroll
                    for (int ks = 0; ks < 4; ++ks) {
                        const int kk0 = ks * 16 + 8 * (g4 >> 1) + q, kk1 = kk0 + 4;
                        const int ga0 = (kk0 >> 3) * LPW + (kk0 & 7), ga1 = (kk1 >> 3) * LPW + (kk1 & 7);
                        const int gb0 = (2 * (kk0 >> 3)) * HPW + 2 * (kk0 & 7), gb1 = (2 * (kk1 >> 3)) * HPW + 2 * (kk1 & 7);
                        const Frag<T> af = frag_tr16<T>(gpatch + ga0 * LP + acol, gpatch + ga1 * LP + acol);
                        const Frag<T> bf = frag_tr16<T>(apatch + (gb0 + toff) * HP + bcol, apatch + (gb1 + toff) * HP + bcol);
                        mma(wacc[j], af, bf);
                        if (ks & 1) __builtin_amdgcn_sched_barrier(0);
                    }
                }
            }
        }
        __syncthreads();                                   // (C) accumulator tile complete
        // ---- epilogue in chunk layout: 1024 chunks (16 x 16 pixels x 4 channel quarters), 2 per thread; the tile leaves as 16
        // rows of 1 KiB
        if (!VAE_ABLATE(a.ablate, 2)) {
            const int qq = tid & 3;
            f32x2 esc2[4], esh2[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) { esc2[e] = *reinterpret_cast<const f32x2*>(cfa + qq * 8 + 2 * e); esh2[e] = *reinterpret_cast<const f32x2*>(cfa + 32 + qq * 8 + 2 * e); }
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int pix = (tid + 512 * u) >> 2;
                const T8 yv8 = *reinterpret_cast<const T8*>(ytile + pix * HP + qq * 16);
                const f32x4 a0_ = *reinterpret_cast<const f32x4*>(ftile + pix * FP + qq * 32), a1_ = *reinterpret_cast<const f32x4*>(ftile + pix * FP + qq * 32 + 16);
                T8 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const f32x2 yv = {(float)yv8[2 * e], (float)yv8[2 * e + 1]};
                    const f32x2 z = yv * esc2[e] + esh2[e];
                    f32x2 g = e < 2 ? f32x2{a0_[2 * e], a0_[2 * e + 1]} : f32x2{a1_[2 * e - 4], a1_[2 * e - 3]};
                    g.x = z.x > 0.f ? g.x : g.x * a.slope; g.y = z.y > 0.f ? g.y : g.y * a.slope;
                    T o0, o1;
                    const f32x2 dzv = round_pair<T>(g, o0, o1);
                    o[2 * e] = o0; o[2 * e + 1] = o1;
                    s1[e] += dzv; s2[e] += dzv * yv;
                }
                const uint32_t off = (uint32_t)(((b * Hg + 2 * y0 + (pix >> 4)) * Wg + 2 * x0 + (pix & 15)) * 32 + qq * 8) * 2u;
                *reinterpret_cast<T8*>(at_bytes(a.dzprev, off)) = o;
            }
        }
    }

    // ---- workgroup results: statistics (every thread holds partial sums of its 8 channels: lanes with equal tid & 3 share
    // them), weight-gradient slab
    float* redw = reinterpret_cast<float*>(ftile);        // [8 waves][32][2] (the accumulator tile is free now)
    __syncthreads();
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        float v[4] = {s1[e].x, s1[e].y, s2[e].x, s2[e].y};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
#pragma unroll
            for (int o = 4; o < 64; o <<= 1) v[k] += __shfl_xor(v[k], o, 64);
        }
        if (lane < 4) {
            const int c = lane * 8 + 2 * e;
            redw[(wave * 32 + c) * 2] = v[0]; redw[(wave * 32 + c + 1) * 2] = v[1];
            redw[(wave * 32 + c) * 2 + 1] = v[2]; redw[(wave * 32 + c + 1) * 2 + 1] = v[3];
        }
    }
    if (wave >= 2) {
#pragma unroll
        for (int j = 0; j < NWT; ++j) {
            const int idx = wj + 6 * j;
            {
                const int tp = idx >> 1, cib = idx & 1;
#pragma unroll
                for (int i = 0; i < 16; ++i)
                    a.slab[(((size_t)blockIdx.x * 9 + tp) * CLO + cib * 32 + acc_row(i, lane)) * 32 + r] = wacc[j][i];
            }
        }
    }
    __syncthreads();
    if (tid < 32) {
        float v1 = 0.f, v2 = 0.f;
#pragma unroll
        for (int w = 0; w < 8; ++w) { v1 += redw[(w * 32 + tid) * 2]; v2 += redw[(w * 32 + tid) * 2 + 1]; }
        v2 = a.ocoef[LC_INVSTD * 32 + tid] * v2 + a.ocoef[LC_XM * 32 + tid] * v1;   // sum dz*xhat from sum dz*y and sum dz
        double* st_ = a.stat + stat_rep() * 64;
        unsafeAtomicAdd(&st_[tid], (double)v1);
        unsafeAtomicAdd(&st_[32 + tid], (double)v2);
    }
}

static inline size_t conv_fused_lds() {
    return (size_t)81 * (64 * 2 + 16) + (size_t)289 * 80 + (size_t)256 * 80 + (size_t)9 * 8 * 32 * 16 + (size_t)(3 * 64 + 64 + 4 * 32 * 2) * 4 + (size_t)5 * 512 * 8 + (size_t)256 * (32 * 4 + 16);
}
