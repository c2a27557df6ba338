// Output conv of the VAE (final_layer.3: Conv2d(32 -> 1, 3x3, pad 1) + sigmoid + BCE) - forward, loss, input gradient, weight
// gradient and the BatchNorm-backward statistics of final_layer.1 in ONE streaming pass over y7, for 128-pixel-wide images and
// 16-bit storage (gfx950).  Same arithmetic, element for element, as convout_step_mfma_kernel (conv_mfma.cuh); what changes
// is the walk:
//
//   * a workgroup (1024 threads = 16 waves, one per CU) streams whole image rows top to bottom through LDS rings instead of
//     cutting the image into 8x32 tiles with a 2-pixel halo: y7 is staged x1.0 (+2 rows per band) instead of x1.69 and the
//     logits are computed x1.0 instead of x1.33;
//   * every MFMA runs "transposed" (the 32 pixels of a block are the N dimension): the nine tap products of a pixel land in
//     the registers of ITS lane (5 LDS writes, conflict-free), and the input gradient of a pixel lands as 4 x 4 consecutive
//     channels in its lane, so the epilogue reads y and writes dz as 8-byte LDS accesses and works on channel pairs with
//     packed f32 math (the tile kernel: one 2-byte LDS read and write and ~12 scalar operations per element); the second
//     BatchNorm-backward statistic is accumulated as sum dz*y and turned into sum dz*xhat7 once per workgroup, in f64;
//   * y7 and the targets never pass through registers on their way in: LDS-DMA (global_load_lds) copies them into LDS rings
//     two ticks ahead; the swizzle of the y ring is applied on the source side (lane i of a copy fetches the chunk that belongs
//     at linear position i).  The copies are invisible to the compiler's vmcnt bookkeeping: the issuing wave waits for the
//     copies of the next tick itself (they complete in issue order) before the tick's closing barrier;
//   * a tick = 2 image rows = 8 blocks of 32 pixels and has two phases separated by raw s_barriers; the waves are split into
//     two groups with their own code (and register allocation), one wave of each kind per SIMD and phase:
//       phase 1  group B (waves 8..15): stage rows s, s+1 (y ring -> BatchNorm + LeakyReLU -> a ring);
//                group A (waves 0..3):  logits / sigmoid / BCE / dlogit of rows s-3, s-2 (tap products of rows s-4 .. s-1
//                                       are in LDS since the last tick);
//       phase 2  group B: issue the copies of tick +2; tap products of rows s, s+1; weight gradient of rows s-4, s-3;
//                group A (waves 0..7): input gradient, epilogue and dz store of rows s-4, s-3.
//     Nothing inside a phase depends on another wave's work of the same phase; four waves per SIMD hide the LDS / MFMA /
//     transcendental latencies of one another (with 8 waves doing everything the kernel ran at ~45 % VALU issue).
// LDS (155 KiB): y ring 10 rows x 8 KiB and a ring 6 rows x 8 KiB (16-byte chunks XOR-swizzled by (pixel >> 2) & 3: conflict-
// free b128 fragment reads, b64 transposed reads and b64 epilogue accesses), tap products [4 rows][9][136] f32, dlogit [4 rows]
// [3 shifted copies][144] 16-bit, every row stored twice (slots q and q + 4: a consumer's rows base, base-1, base-2 never wrap, so
// its addresses are a per-lane constant plus one scalar; a weight-gradient B fragment is one aligned b128 read), targets [8 rows][128] f32.  Ring
// slots follow a running tick counter, not the row number, so consecutive units of a workgroup never collide.
#pragma once
#include "conv_mfma.cuh"
#include "conv_deep.cuh"

template <typename T> struct ConvOutStreamArgs {
    const T* yf; const float* wt; const float* bias; const float* target;
    float* xhat; double* accum;
    T* dz; float* slab; double* stat;
    int B, H, RB, nb, n_units; float inv_n, slope, gmul;   // RB rows per band, nb bands per image, n_units = B * nb
    BnFuse fuse;
    long long* dbg;
};

namespace cos {
static constexpr int RW = 128, NA = 6, NY = 10, ROWB = RW * 64, PPW = 136, DLW = 144, NTG = 8, RED = 32 * 9 + 64 + 2;
__device__ __forceinline__ int ring_off(int px, int chunk) { return px * 64 + ((chunk ^ ((px >> 2) & 3)) << 4); }
// LDS-DMA, 4 / 16 B per lane (same M0 protocol as deep::dma16).  No "memory" clobber on purpose: the copies read tensors no
// kernel writes while this one runs and fill ring slots that no compiler-visible access of the same phase touches (the phase
// barriers, which do clobber memory, order them against the consumers), so the compiler may move LDS reads across them.
__device__ __forceinline__ void dma4(const void* gsrc, char* lds_dst) {
    unsigned keep;
    const unsigned l = (unsigned)(unsigned long long)(__attribute__((address_space(3))) char*)lds_dst;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(l));
}
__device__ __forceinline__ void dma16(const void* gsrc, char* lds_dst) {
    unsigned keep;
    const unsigned l = (unsigned)(unsigned long long)(__attribute__((address_space(3))) char*)lds_dst;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(l));
}
// at most n of this wave's vector-memory operations still outstanding (they complete in issue order)
__device__ __forceinline__ void wait_vm(int n) {
    switch (n) {
        case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
        case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
        case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
        case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
        case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
        case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
        case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
        case 7: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
        default: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
    }
}
}
static inline size_t convout_stream_lds() {
    return (size_t)(cos::NY + cos::NA) * cos::ROWB + 4 * 9 * cos::PPW * 4 + 8 * 3 * cos::DLW * 2 + cos::NTG * cos::RW * 4 + 128 * 4;
}

template <typename T>
__global__ __launch_bounds__(1024) void convout_stream_kernel(ConvOutStreamArgs<T> a) {
    using namespace cos;
    typedef typename H16<T>::v8 T8;
    typedef __attribute__((ext_vector_type(4))) T T4;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* yring = smem;                                                    // raw y (dz in place), filled by LDS-DMA
    char* aring = yring + NY * ROWB;                                       // a = LeakyReLU(BN(y))
    float* part = reinterpret_cast<float*>(aring + NA * ROWB);             // [4][9][PPW], pixel x at index x + 4
    T* dlc = reinterpret_cast<T*>(part + 4 * 9 * PPW);                     // [8][3][DLW], copy c at index i + 8 holds dl[i - c + 1]; row q also at q + 4
    float* tgr = reinterpret_cast<float*>(dlc + 8 * 3 * DLW);              // [NTG][RW] targets, filled by LDS-DMA
    float* cf = tgr + NTG * RW;                                            // scale | shift | invstd | -mean*invstd
    float (*red)[RED] = reinterpret_cast<float (*)[RED]>(part);            // final reductions (the tap products are dead by then)
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), r = lane & 31, h = lane >> 5;
    const int H = a.H, G = gridDim.x, K = a.RB / 2 + 3;
    const int wq = wave & 7, brow = wq >> 2, x0 = (wq & 3) * 32;          // phase 2: this wave's block (group B also: the quarter row its copies fill)
#ifdef VAE_PHASE_STAMPS
    long long dst_[8] = {0, 0, 0, 0, 0, 0, 0, 0}; const long long dst_entry_ = clock64(); long long dst_t_ = dst_entry_;
#define CSTAMP(k) { __builtin_amdgcn_sched_barrier(0); const long long t1_ = clock64(); dst_[k] += t1_ - dst_t_; dst_t_ = t1_; __builtin_amdgcn_sched_barrier(0); }
#else
#define CSTAMP(k)
#endif

    if (tid < 32) { float k1; bn_fused_channel(a.fuse, tid, blockIdx.x == 0, cf[tid], k1, cf[32 + tid], &cf[64 + tid], &cf[96 + tid]); }
    for (int i = tid; i < 4 * 9 * PPW; i += 1024) part[i] = 0.f;
    for (int i = tid; i < 8 * 3 * DLW / 2; i += 1024) reinterpret_cast<int*>(dlc)[i] = 0;

    if (wave >= 8) {
        // ====================================== group B: copies, staging, tap products, weight gradient ======================================
        Frag<T> wfA[2];      // tap products: A[m = tap r][k = channel]
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int j = 0; j < 8; ++j) wfA[ks].v[j] = (T)(r < 9 ? a.wt[r * 32 + ks * 16 + 8 * h + j] : 0.f);
        f32x16 accw;
#pragma unroll
        for (int i = 0; i < 16; ++i) accw[i] = 0.f;
        const int g4 = lane >> 4, li = lane & 15, q = li >> 2, p = li & 3;
        // per-lane address constants (bytes): a tick adds one scalar to each
        // dW's B fragment: 8 pixels x0 + 8h .. of copy kx of dl row (base - ky), tap = min(r, 8) (columns >= 9 of dW are never read)
        const int tW = r < 9 ? r : 8, kyW = tW / 3, kxW = tW - 3 * kyW;
        const int offW = ((-kyW * 3 + kxW) * DLW + x0 + 8 * h + 8) * 2;
        // dW's A fragment (transposed reads of the a ring) and the tap products' b128 cells
        const int pxa = x0 + 8 * (g4 >> 1) + q, chA = 2 * (g4 & 1) + (p >> 1);
        const int offT0 = ring_off(pxa, chA) + (p & 1) * 8, offT1 = ring_off(pxa + 4, chA) + (p & 1) * 8;   // (+1024: the second k-step, same swizzle)
        const int offB0 = ring_off(x0 + r, h), offB1 = ring_off(x0 + r, 2 + h);                             // channels 8h.., 16 + 8h..
        const int offP = ((4 * h) * PPW + 4 + x0 + r) * 4;                                                  // tap products of taps 4h ..
        // LDS-DMA of one tick: y rows s, s+1 of image b into y-ring slots ys, ys+1 (wave: row `brow`, quarter `wq & 3`, two 1 KiB
        // copies; lane i of a copy fetches the chunk stored at linear position i), targets of logit rows s-3, s-2 into target slots
        // ts0, ts0+1 (waves 8..11: row wq >> 1, half wq & 1).  Rows outside the image or the band's needs copy row 0 (never used).
        const int dci = (wq & 3) * 128 + lane, dpx = dci >> 2;
        const int dsrc0 = dpx * 64 + (((dci & 3) ^ ((dpx >> 2) & 3)) << 4);            // source byte offset in the row, first copy (second: + 1024)
        int ua = blockIdx.x, ka = 0, ya = 0, ta = 5;        // the tick two ahead: unit, tick, y slot of its row s, target slot of its logit row s-3
        int ba = 0, r0a = 0;
        if (ua < a.n_units) { ba = ua / a.nb; r0a = (ua - ba * a.nb) * a.RB; }
        const int nD = wq < 4 ? 3 : 2;                      // copies per tick of this wave
        auto issue_ahead = [&]() __attribute__((always_inline)) {   // returns the number of copies this wave issued
            const bool live = ua < a.n_units;
            if (live) {
                const int s = r0a - 2 + 2 * ka, r1 = r0a + a.RB;
                if (wq < 4) {
                    const int row = s - 3 + (wq >> 1);
                    const bool ok = row >= 0 && row < H && row >= r0a - 1 && row <= r1;
                    const float* src = a.target + ((size_t)(ba * H + (ok ? row : 0))) * RW + (wq & 1) * 64 + lane;
                    const int ts = (ta + (wq >> 1)) & (NTG - 1);
                    dma4(src, reinterpret_cast<char*>(tgr + ts * RW + (wq & 1) * 64));
                }
                const int row = s + brow;
                const bool ok = row >= 0 && row < H && row <= r1 + 1;
                const char* rowp = reinterpret_cast<const char*>(a.yf + ((size_t)(ba * H + (ok ? row : 0)) * RW) * 32) + dsrc0;
                int slot = ya + brow; slot = slot >= NY ? slot - NY : slot;
                char* dst = yring + slot * ROWB + (wq & 3) * 2048;
                dma16(rowp, dst);
                dma16(rowp + 1024, dst + 1024);
            }
            ya = ya + 2 >= NY ? ya + 2 - NY : ya + 2; ta = (ta + 2) & (NTG - 1);
            if (++ka == K) {
                ka = 0; ua += G;
                if (ua < a.n_units) { ba = ua / a.nb; r0a = (ua - ba * a.nb) * a.RB; }
            }
            return live ? nD : 0;
        };
        issue_ahead();
        const int nd1 = issue_ahead();
        deep::barrier_lds();                 // cf published, borders zeroed
        f32x2 kc[4], kh[4];                  // staging: the 8 channels of this thread's two chunks (linear position st of rows s, s + 1)
        const int st = tid & 511, spos = st * 16;
        {
            const int px = st >> 2, ch = (st & 3) ^ ((px >> 2) & 3);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                kc[e] = f32x2{cf[ch * 8 + 2 * e], cf[ch * 8 + 2 * e + 1]};
                kh[e] = f32x2{cf[32 + ch * 8 + 2 * e], cf[32 + ch * 8 + 2 * e + 1]};
            }
        }
        wait_vm(nd1);                        // the first tick's copies (issued first) landed; the second tick's may be outstanding
        deep::barrier_lds();
        CSTAMP(0)
        int py = 0, pa = 0, p4 = 0;          // ring positions of row s (y, a, tap products)
        for (int unit = blockIdx.x; unit < a.n_units; unit += G) {
            const int b = unit / a.nb, r0 = (unit - b * a.nb) * a.RB, r1 = r0 + a.RB;
            for (int k = 0; k < K; ++k) {
                const int s = r0 - 2 + 2 * k;
                // ---- phase 1: stage rows s, s + 1: y ring -> BatchNorm + LeakyReLU -> a ring
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int row = s + u;
                    const bool ok = row >= 0 && row < H && row <= r1 + 1;
                    int ys = py + u; ys = ys >= NY ? ys - NY : ys;
                    int as = pa + u; as = as >= NA ? as - NA : as;
                    char* adst = aring + as * ROWB + spos;
                    if (!ok) { *reinterpret_cast<T8*>(adst) = T8{0, 0, 0, 0, 0, 0, 0, 0}; continue; }   // (wave-uniform) outside the image: a = 0
                    const T8 yv = *reinterpret_cast<const T8*>(yring + ys * ROWB + spos);
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
                CSTAMP(1)
                deep::barrier_lds();
                CSTAMP(2)
                // ---- phase 2: copies of tick + 2, tap products of a row s + brow, weight gradient of row s - 4 + brow
                const int nd = issue_ahead();
                {
                    int as = pa + brow; as = as >= NA ? as - NA : as;
                    const char* arow_ = aring + as * ROWB;
                    f32x16 acc;
#pragma unroll
                    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
                    Frag<T> b0 = load_frag(reinterpret_cast<const T*>(arow_ + offB0)), b1 = load_frag(reinterpret_cast<const T*>(arow_ + offB1));
                    mma(acc, wfA[0], b0);
                    mma(acc, wfA[1], b1);
                    float* pr = reinterpret_cast<float*>(reinterpret_cast<char*>(part + ((p4 + brow) & 3) * 9 * PPW) + offP);
                    pr[0] = acc[0]; pr[PPW] = acc[1]; pr[2 * PPW] = acc[2]; pr[3 * PPW] = acc[3];     // taps 4h .. 4h + 3
                    if (h == 0) pr[8 * PPW] = acc[4];                                                 // tap 8
                }
                const int Rf = s - 4 + brow;
                if (Rf >= r0 && Rf < r1) {
                    // dW[channel][tap] += sum_pixels a[pixel][channel] dl[pixel - tap]; dl row Rf + 1 (tap row 0) at mirrored slot 4 + ((p4 + 1 + brow) & 3)
                    int as = pa + 2 + brow; as = as >= NA ? as - NA : as;            // a ring slot of row s - 4 + brow
                    const char* arow_ = aring + as * ROWB;
                    const char* dbase = reinterpret_cast<const char*>(dlc + (4 + ((p4 + 1 + brow) & 3)) * 3 * DLW);
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks) {
                        Frag<T> afr = frag_tr16<T>(arow_ + offT0 + ks * 1024, arow_ + offT1 + ks * 1024);
                        Frag<T> bfr = load_frag(reinterpret_cast<const T*>(dbase + offW + ks * 32));
                        mma(accw, afr, bfr);
                    }
                }
                CSTAMP(3)
                wait_vm(nd);                 // the next tick's copies landed (only this tick's, issued after them, may be outstanding)
                deep::barrier_lds();
                CSTAMP(4)
                py = py + 2 >= NY ? py + 2 - NY : py + 2; pa = pa + 2 >= NA ? pa + 2 - NA : pa + 2; p4 = (p4 + 2) & 3;
            }
        }
        if (r < 9) {
#pragma unroll
            for (int i = 0; i < 16; ++i) red[wq][r * 32 + acc_row(i, lane)] = accw[i];
        }
    } else {
        // ====================================== group A: logits (waves 0..3), input gradient + epilogue (waves 0..7) ======================================
        Frag<T> wfT;         // input gradient: A[m = channel r][k = tap]
#pragma unroll
        for (int j = 0; j < 8; ++j) { const int t = 8 * h + j; wfT.v[j] = (T)(t < 9 ? a.wt[t * 32 + r] : 0.f); }
        const float bo = a.bias[0], gs = a.gmul;
        float bsum = 0.f, sdl = 0.f;
        f32x2 s1[8], s2[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) { s1[e] = f32x2{0.f, 0.f}; s2[e] = f32x2{0.f, 0.f}; }
        const int lrow = wave >> 1, lx = tid & 127;           // phase 1, waves 0..3: this thread's logit pixel
        // dA's B fragment, element j: tap 8h + j of pixel x0 + r from copy 1 of dl row (base - ky); taps >= 9 read a cell that stays zero
        int offA[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int t = 8 * h + j, ky = t / 3, kx = t - 3 * ky;
            offA[j] = t < 9 ? ((-ky * 3 + 1) * DLW + x0 + r - kx + 1 + 8) * 2 : 0;
        }
        const int offE = (x0 + r) * 64 + 8 * h, swz = (((x0 + r) >> 2) & 3) << 4;                            // epilogue cell of chunk g: offE + ((g << 4) ^ swz)
        const int offS0 = ring_off(x0 + (lane >> 2), lane & 3), offS1 = ring_off(x0 + 16 + (lane >> 2), lane & 3);   // dz store
        deep::barrier_lds();                 // cf published, borders zeroed
        // epilogue: the lane's 16 channels, pair e = channels acc_row(2e, lane), +1
        f32x2 esc[8], esh[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int c = acc_row(2 * e, lane);
            esc[e] = f32x2{cf[c], cf[c + 1]}; esh[e] = f32x2{cf[32 + c], cf[33 + c]};
        }
        deep::barrier_lds();
        CSTAMP(0)
        int py = 0, pa = 0, p4 = 0, p8 = 5;  // ring positions of row s (y, a, tap products; dlogit row s-3 = p4 + 1), of target row s-3
        for (int unit = blockIdx.x; unit < a.n_units; unit += G) {
            const int b = unit / a.nb, r0 = (unit - b * a.nb) * a.RB, r1 = r0 + a.RB;
            for (int k = 0; k < K; ++k) {
                const int s = r0 - 2 + 2 * k;
                // ---- phase 1 (waves 0..3): logits / sigmoid / BCE / dlogit of rows s - 3, s - 2 (a wave = half a row)
                if (wave < 4) {
                    const int R = s - 3 + lrow;
                    const bool ok = R >= 0 && R < H && R >= r0 - 1 && R <= r1;
                    const float tg = tgr[((p8 + lrow) & (NTG - 1)) * RW + lx];
                    float logit = bo;
#pragma unroll
                    for (int t = 0; t < 9; ++t) logit += part[(((p4 + lrow + t / 3) & 3) * 9 + t) * PPW + 4 + lx + t % 3 - 1];
                    const float xh = 1.f / (1.f + expf(-logit));
                    const float om = xh * (1.f - xh);
                    const float dlv = (xh - tg) / fmaxf(om, 1e-12f) * om * a.inv_n;
                    const float dl = ok ? dlv * gs : 0.f;
                    const T dlt = (T)dl;
                    T* drow = dlc + ((p4 + 1 + lrow) & 3) * 3 * DLW + lx + 8;
                    drow[-1] = dlt; drow[DLW] = dlt; drow[2 * DLW + 1] = dlt;
                    drow[12 * DLW - 1] = dlt; drow[13 * DLW] = dlt; drow[14 * DLW + 1] = dlt;      // the same row at slot + 4
                    if (ok && R >= r0 && R < r1) {        // the band's own rows
                        const float l1 = fmaxf(logf(xh), -100.f), l0 = fmaxf(logf(1.f - xh), -100.f);
                        bsum += -(tg * l1 + (1.f - tg) * l0);
                        a.xhat[((size_t)(b * H + R)) * RW + lx] = xh;
                        sdl += dl;
                    }
                }
                CSTAMP(1)
                deep::barrier_lds();
                CSTAMP(2)
                // ---- phase 2: input gradient, epilogue and dz store of row s - 4 + brow
                const int Rf = s - 4 + brow;
                if (Rf >= r0 && Rf < r1) {
                    int ys = py + NY - 4 + brow; ys = ys >= NY ? ys - NY : ys;       // y ring slot of row s - 4 + brow
                    char* yrow = yring + ys * ROWB;
                    // dl row Rf + 1 (tap row 0) sits at logical slot p4 + 1 + brow; its mirrored slot in 4..7 never wraps going down
                    const char* dbase = reinterpret_cast<const char*>(dlc + (4 + ((p4 + 1 + brow) & 3)) * 3 * DLW);
                    T4 yq[4];
#pragma unroll
                    for (int g = 0; g < 4; ++g) yq[g] = *reinterpret_cast<const T4*>(yrow + offE + ((g << 4) ^ swz));
                    // dA[channel][pixel] = sum_t w[t][channel] dl[pixel - t]
                    f32x16 acca;
#pragma unroll
                    for (int i = 0; i < 16; ++i) acca[i] = 0.f;
                    {
                        Frag<T> bf;
#pragma unroll
                        for (int j = 0; j < 8; ++j) bf.v[j] = *reinterpret_cast<const T*>(dbase + offA[j]);
                        mma(acca, wfT, bf);
                    }
                    // epilogue: dz = dA * leaky'(z), in place over y (this lane: pixel x0 + r, channels 8g + 4h .. + 3)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        T4 o4;
#pragma unroll
                        for (int e2 = 0; e2 < 2; ++e2) {
                            const int e = 2 * g + e2;
                            const f32x2 yv = f32x2{(float)yq[g][2 * e2], (float)yq[g][2 * e2 + 1]};
                            const f32x2 z = f32x2{__builtin_fmaf(yv.x, esc[e].x, esh[e].x), __builtin_fmaf(yv.y, esc[e].y, esh[e].y)};
                            const float d0 = acca[4 * g + 2 * e2], d1 = acca[4 * g + 2 * e2 + 1];
                            const T o0 = (T)(z.x > 0.f ? d0 : d0 * a.slope), o1 = (T)(z.y > 0.f ? d1 : d1 * a.slope);
                            const f32x2 dzv = f32x2{(float)o0, (float)o1};
                            s1[e] += dzv;
                            s2[e] = f32x2{__builtin_fmaf(dzv.x, yv.x, s2[e].x), __builtin_fmaf(dzv.y, yv.y, s2[e].y)};   // sum dz*y (see the reduction)
                            o4[2 * e2] = o0; o4[2 * e2 + 1] = o1;
                        }
                        *reinterpret_cast<T4*>(yrow + offE + ((g << 4) ^ swz)) = o4;
                    }
                    // (the wave re-reads only its own 32 pixels: LDS executes a wave's accesses in order, no wait needed)
                    char* dg = reinterpret_cast<char*>(a.dz + ((size_t)(b * H + Rf) * RW + x0) * 32) + lane * 16;
                    const T8 v0 = *reinterpret_cast<const T8*>(yrow + offS0), v1 = *reinterpret_cast<const T8*>(yrow + offS1);
                    *reinterpret_cast<T8*>(dg) = v0;
                    *reinterpret_cast<T8*>(dg + 1024) = v1;
                }
                CSTAMP(3)
                deep::barrier_lds();
                CSTAMP(4)
                py = py + 2 >= NY ? py + 2 - NY : py + 2; pa = pa + 2 >= NA ? pa + 2 - NA : pa + 2; p4 = (p4 + 2) & 3; p8 = (p8 + 2) & (NTG - 1);
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
        sdl = wave_sum(sdl); bsum = wave_sum(bsum);
        if (r == 0) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int c = acc_row(2 * e, lane);
                red[wave][288 + c] = s1[e].x; red[wave][288 + c + 1] = s1[e].y;
                red[wave][320 + c] = s2[e].x; red[wave][320 + c + 1] = s2[e].y;
            }
        }
        if (lane == 0) { red[wave][352] = sdl; red[wave][353] = bsum; }
    }
    // ---- workgroup reductions: dW (rows = channel, lanes 0..8 = tap) from group B, statistics / sum of dlogit / BCE sum from group A
    deep::barrier_lds();
    auto rsum = [&](int j) { float v = 0.f;
#pragma unroll
        for (int w = 0; w < 8; ++w) v += red[w][j];
        return v; };
    if (tid < 288) a.slab[(size_t)blockIdx.x * 288 + tid] = rsum(tid);
    else if (tid < 288 + 32) unsafeAtomicAdd(&a.stat[stat_rep() * 64 + tid - 288], (double)rsum(tid));
    else if (tid < 288 + 64) {   // sum dz*xhat7 with xhat7 = y*invstd - mean*invstd, from this workgroup's sums of dz*y and dz
        const int c = tid - 320;
        unsafeAtomicAdd(&a.stat[stat_rep() * 64 + 32 + c], (double)cf[64 + c] * (double)rsum(tid) + (double)cf[96 + c] * (double)rsum(288 + c));
    }
    else if (tid == 352) unsafeAtomicAdd(&a.accum[stat_rep() * 8 + 2], (double)rsum(352));
    else if (tid == 353) unsafeAtomicAdd(&a.accum[stat_rep() * 8 + 0], (double)rsum(353));
#ifdef VAE_PHASE_STAMPS
    if (a.dbg && lane == 0) { dst_[5] = clock64() - dst_entry_; for (int k_ = 0; k_ < 8; ++k_) a.dbg[((size_t)blockIdx.x * 16 + wave) * 8 + k_] = dst_[k_]; }
#endif
#undef CSTAMP
}
