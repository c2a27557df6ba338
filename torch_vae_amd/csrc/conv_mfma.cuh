// Implicit-GEMM 3x3 stride-2 convolution kernels on MFMA for gfx950.
//
// Internal activation layout is NHWC ("channels-last"): the MFMA A operand needs
// 8 consecutive k (= input channels) per lane, which NHWC gives as one 16-byte
// access.  Four kernels cover every conv / transposed-conv of the VAE:
//
//   down_kernel  out[b,oy,ox,n] = sum_{ky,kx,k} f(in[b,2oy+ky-1,2ox+kx-1,k]) * Wp[ky*3+kx][k][n]
//                = Conv2d(k3,s2,p1) forward          (models.py:45)   and
//                = ConvTranspose2d input-gradient    (autograd of models.py:66-68,77)
//   up_kernel    out[b,oy,ox,n] = sum_{ky,kx: (oy+1-ky),(ox+1-kx) even} f(in[b,(oy+1-ky)/2,(ox+1-kx)/2,k]) * Wp[..][k][n]
//                = ConvTranspose2d(k3,s2,p1,op1) forward (models.py:66-68,77) and
//                = Conv2d input-gradient
//   wgrad_kernel dW[a][b][ky][kx] = sum_{img,y,x} fS(S[img,y,x,a]) * fG(G[img,2y+ky-1,2x+kx-1,b])
//                = weight gradient of both layer kinds (S = low-res side, G = high-res side)
//   dense_kernel C[m][n] = sum_k f(A[m][k]) * Bp[k][n]   (fc_mu|fc_var forward, decoder_input dgrad)
//
// f() is the per-channel affine+LeakyReLU load transform of common.cuh: the
// producer's train-mode BatchNorm + LeakyReLU (forward operands) or the BatchNorm
// backward (gradient operands) is applied while the tile is staged into LDS, so
// neither the normalised activation nor dL/dy is ever materialised in HBM.
// Epilogues emit the per-channel sums train-mode BatchNorm needs (H2 in SURVEY.md 7).
#pragma once
#include "common.cuh"

enum { EPI_FWD = 0, EPI_BWD = 1, EPI_PLAIN = 2 };

template <typename T> struct ConvArgs {
    const T* src0; const T* src1; const float* coef; float slope;  // input + load transform (rows p0,p1,p2, stride Cin)
    const T* wp; const float* bias;                                // packed [9][Cin/8][Cout][8]
    T* out;
    const T* yout; const float* ocoef; float oslope;               // EPI_BWD: output-side layer block (rows LC_*, stride Cout)
    double* stat;                                                  // [2][Cout] (EPI_FWD: sum y, sum y^2; EPI_BWD: sum dz, sum dz*xhat)
    int B, Hs, Ws, Cin, Cout;                                      // Hs,Ws: low-res side (down: output, up: input)
    int lth, ltw, lTB, tiles_x, tiles_y;
    int two_src, epi;                                              // runtime: gradient-operand load / epilogue kind
    unsigned m_pp, m_pw, m_tx, m_txy;                              // fastdiv magics: PP, PW, tiles_x, tiles_x*tiles_y
    long long* dbg;                                                // diagnostic builds only: per-wave phase cycle counters
    BnFuse fuse;                                                   // mode != 0: derive the staging coefficients from batch statistics (pipelined kernels)
    int rev, n_mt;                                                 // walk the M tiles in reverse order (n_mt of them)
    int xcd;                                                       // gridDim/8 when the persistent kernels renumber their workgroups per XCD (0: off)
    // pipelined kernels: when set, every staged (transformed, storage-rounded) chunk a tile OWNS (its non-halo pixels, N tile 0)
    // is also written here, in the source tensor's layout: the materialised activation / BatchNorm-backward gradient that the
    // deep layers' weight-gradient kernels then read without any staging arithmetic
    T* stage_out;
};

// x / d for small x via one mul_hi: m = ceil(2^32 / d), exact for x, d < 2^16
__device__ __forceinline__ int fastdiv(int x, unsigned m) { return m ? (int)__umulhi((unsigned)x, m) : x; }  // m == 0 encodes d == 1
static inline unsigned fastdiv_magic(int d) { return d <= 1 ? 0u : (unsigned)((0x100000000ULL + (unsigned long long)d - 1) / (unsigned long long)d); }

static constexpr int PATCH_PITCH = 80;  // bytes per staged pixel: 64 B of channels + 16 B pad (LDS bank spread)

template <typename T>
__device__ __forceinline__ Vec16<T> load_transform16(const T* s0, const T* s1, bool two, size_t g, const float* cf,
                                                     int C, int cb, float slope) {
    Vec16<T> v0 = *reinterpret_cast<const Vec16<T>*>(s0 + g);
    Vec16<T> o;
    if (two) {
        Vec16<T> v1 = *reinterpret_cast<const Vec16<T>*>(s1 + g);
#pragma unroll
        for (int e = 0; e < Vec16<T>::N; ++e)
            o.set(e, leaky(v0.get(e) * cf[cb + e] + v1.get(e) * cf[C + cb + e] + cf[2 * C + cb + e], slope));
    } else {
#pragma unroll
        for (int e = 0; e < Vec16<T>::N; ++e) o.set(e, leaky(v0.get(e) * cf[cb + e] + cf[2 * C + cb + e], slope));
    }
    return o;
}


template <typename T>
__device__ __forceinline__ Vec16<T> transform16(const Vec16<T>& v0, const Vec16<T>& v1, bool two, const float* cf, int C,
                                                int cb, float slope) {
    Vec16<T> o;
    if (two) {
        // gradient operand: dL/dy = dz*p0 + y*p1 + p2 (the launchers pass slope = 1, so no LeakyReLU here)
#pragma unroll
        for (int e = 0; e < Vec16<T>::N; ++e)
            o.set(e, v0.get(e) * cf[cb + e] + v1.get(e) * cf[C + cb + e] + cf[2 * C + cb + e]);
    } else {
#pragma unroll
        for (int e = 0; e < Vec16<T>::N; ++e) o.set(e, leaky(v0.get(e) * cf[cb + e] + cf[2 * C + cb + e], slope));
    }
    return o;
}

// Stage `nitems` 16-byte chunks global -> (transform) -> LDS.  NB chunks per thread are loaded
// before any is consumed, so NB (x2 for gradient operands) HBM requests per lane are in flight
// instead of one dependent load at a time.  map(it, ok, g, loff, cb): item -> validity, global
// element offset, LDS byte offset, first channel.  Invalid items store zeros (conv padding).
template <typename T, int NB, typename F>
__device__ __forceinline__ void stage_items(int tid, int nitems, const T* s0, const T* s1, bool two, const float* cf,
                                            int C, float slope, char* lds, F&& map) {
    for (int base = tid; base < nitems; base += 256 * NB) {
        Vec16<T> v0[NB], v1[NB];
        int loff[NB], cb[NB];
        bool ok[NB];
#pragma unroll
        for (int u = 0; u < NB; ++u) {
            const int it = base + u * 256;
            size_t g = 0;
            loff[u] = -1; cb[u] = 0; ok[u] = false;
            if (it < nitems) map(it, ok[u], g, loff[u], cb[u]);
            if (!ok[u]) g = 0;
            v0[u] = *reinterpret_cast<const Vec16<T>*>(s0 + g);
            if (two) v1[u] = *reinterpret_cast<const Vec16<T>*>(s1 + g); else v1[u] = v0[u];
        }
#pragma unroll
        for (int u = 0; u < NB; ++u) {
            if (loff[u] >= 0) {
                Vec16<T> o = transform16<T>(v0[u], v1[u], two, cf, C, cb[u], slope);
                if (!ok[u]) o = zero_vec16<T>();
                *reinterpret_cast<Vec16<T>*>(lds + loff[u]) = o;
            }
        }
    }
}

// Common epilogue for one accumulator value.
template <typename T>
__device__ __forceinline__ void epi_store(const ConvArgs<T>& a, size_t idx, int n, float accv, float bv, float sc,
                                          float sh, float is, float xm, float& s1, float& s2) {
    const int EPI = a.epi;
    if (EPI == EPI_FWD) {
        float v = round_as<T>(accv + bv);
        a.out[idx] = fromfloat<T>(v);
        s1 += v; s2 += v * v;
    } else if (EPI == EPI_BWD) {
        float y = tofloat(a.yout[idx]);
        float z = y * sc + sh;
        float dz = round_as<T>(z > 0.f ? accv : accv * a.oslope);
        a.out[idx] = fromfloat<T>(dz);
        s1 += dz; s2 += dz * (y * is + xm);
    } else {
        a.out[idx] = fromfloat<T>(accv);
    }
}

// ---------------------------------------------------------------------------
template <typename T, int NT>
__global__ __launch_bounds__(256, 2) void down_kernel(ConvArgs<T> a) {
    const bool TWO_SRC = a.two_src != 0; const int EPI = a.epi;
    constexpr int CK = 64 / sizeof(T), KS = CK / 16, E16 = 16 / sizeof(T);
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, h = lane >> 5;
    const int th = 1 << a.lth, tw = 1 << a.ltw, TB = 1 << a.lTB;
    const int PH = 2 * th + 1, PW = 2 * tw + 1, PP = PH * PW, npix = TB * PP;
    const int Hin = 2 * a.Hs, Win = 2 * a.Ws, Cin = a.Cin, Cout = a.Cout;
    float* cf = reinterpret_cast<float*>(smem);
    char* patch = smem + ((3 * Cin * 4 + 15) & ~15);
    float* red = reinterpret_cast<float*>(patch + npix * PATCH_PITCH);

    const int tile = blockIdx.x;
    const int bt = fastdiv(tile, a.m_txy), trem = tile - bt * a.tiles_x * a.tiles_y, ty = fastdiv(trem, a.m_tx), tx = trem - ty * a.tiles_x;
    const int b0 = bt << a.lTB, oy0 = ty << a.lth, ox0 = tx << a.ltw, n0 = blockIdx.y * 32 * NT;

    for (int i = tid; i < 3 * Cin; i += 256) cf[i] = a.coef[i];

    const int R = wave * 32 + r;
    const int pbase = ((R >> (a.lth + a.ltw)) * PH + 2 * ((R >> a.ltw) & (th - 1))) * PW + 2 * (R & (tw - 1));

    f32x16 acc[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[nt][i] = 0.f;

    for (int c0 = 0; c0 < Cin; c0 += CK) {
        __syncthreads();
        stage_items<T, 4>(tid, npix * 4, a.src0, a.src1, TWO_SRC, cf, Cin, a.slope, patch,
                          [&](int it, bool& ok, size_t& g, int& loff, int& cb) {
            const int pix = it >> 2, q = it & 3;
            const int img = fastdiv(pix, a.m_pp), rem = pix - img * PP, py = fastdiv(rem, a.m_pw), px = rem - py * PW;
            const int b = b0 + img, iy = 2 * oy0 - 1 + py, ix = 2 * ox0 - 1 + px;
            loff = pix * PATCH_PITCH + q * 16; cb = c0 + q * E16;
            ok = b < a.B && iy >= 0 && iy < Hin && ix >= 0 && ix < Win;
            g = (((size_t)b * Hin + iy) * Win + ix) * Cin + cb;
        });
        __syncthreads();
#pragma unroll
        for (int t = 0; t < 9; ++t) {
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const T* ap = reinterpret_cast<const T*>(patch + (pbase + (t / 3) * PW + (t % 3)) * PATCH_PITCH + ks * 32) + h * 8;
                Frag<T> af = load_frag(ap);
                const size_t kg = (size_t)t * (Cin >> 3) + ((c0 + ks * 16) >> 3) + h;
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    Frag<T> bf = load_frag(a.wp + (kg * Cout + n0 + nt * 32 + r) * 8);
                    mma(acc[nt], af, bf);
                }
            }
        }
    }

    // epilogue: lane = output channel (column), 16 accumulator rows = pixels
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int n = n0 + nt * 32 + r;
        float bv = 0.f, sc = 0.f, sh = 0.f, is = 0.f, xm = 0.f, s1 = 0.f, s2 = 0.f;
        if (EPI == EPI_FWD) bv = a.bias ? a.bias[n] : 0.f;
        if (EPI == EPI_BWD) {
            sc = a.ocoef[LC_SC * Cout + n]; sh = a.ocoef[LC_SH * Cout + n];
            is = a.ocoef[LC_INVSTD * Cout + n]; xm = a.ocoef[LC_XM * Cout + n];
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int RR = wave * 32 + acc_row(i, lane);
            const int b = b0 + (RR >> (a.lth + a.ltw));
            if (b < a.B) {
                const int oy = oy0 + ((RR >> a.ltw) & (th - 1)), ox = ox0 + (RR & (tw - 1));
                const size_t idx = (((size_t)b * a.Hs + oy) * a.Ws + ox) * Cout + n;
                epi_store<T>(a, idx, n, acc[nt][i], bv, sc, sh, is, xm, s1, s2);
            }
        }
        if (EPI != EPI_PLAIN) {
            s1 += __shfl_xor(s1, 32, 64); s2 += __shfl_xor(s2, 32, 64);
            if (h == 0) { red[((wave * NT + nt) * 32 + r) * 2] = s1; red[((wave * NT + nt) * 32 + r) * 2 + 1] = s2; }
        }
    }
    if (EPI != EPI_PLAIN) {
        __syncthreads();
        if (tid < NT * 32) {
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int w = 0; w < 4; ++w) { s1 += red[((w * NT) * 32 + tid) * 2]; s2 += red[((w * NT) * 32 + tid) * 2 + 1]; }
            double* st_ = a.stat + stat_rep() * 2 * Cout;
            unsafeAtomicAdd(&st_[n0 + tid], (double)s1);
            unsafeAtomicAdd(&st_[Cout + n0 + tid], (double)s2);
        }
    }
}

// ---------------------------------------------------------------------------
template <typename T, int NT>
__global__ __launch_bounds__(256, 2) void up_kernel(ConvArgs<T> a) {
    const bool TWO_SRC = a.two_src != 0; const int EPI = a.epi;
    constexpr int CK = 64 / sizeof(T), KS = CK / 16, E16 = 16 / sizeof(T);
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, h = lane >> 5;
    const int th = 1 << a.lth, tw = 1 << a.ltw, TB = 1 << a.lTB;
    const int PH = th + 1, PW = tw + 1, PP = PH * PW, npix = TB * PP;
    const int Hs = a.Hs, Ws = a.Ws, Cin = a.Cin, Cout = a.Cout;
    float* cf = reinterpret_cast<float*>(smem);
    char* patch = smem + ((3 * Cin * 4 + 15) & ~15);
    float* red = reinterpret_cast<float*>(patch + npix * PATCH_PITCH);

    const int tile = blockIdx.x;
    const int bt = fastdiv(tile, a.m_txy), trem = tile - bt * a.tiles_x * a.tiles_y, ty = fastdiv(trem, a.m_tx), tx = trem - ty * a.tiles_x;
    const int b0 = bt << a.lTB, iy0 = ty << a.lth, ix0 = tx << a.ltw, n0 = blockIdx.y * 32 * NT;

    for (int i = tid; i < 3 * Cin; i += 256) cf[i] = a.coef[i];

    const int R = wave * 32 + r;
    const int pbase = ((R >> (a.lth + a.ltw)) * PH + ((R >> a.ltw) & (th - 1))) * PW + (R & (tw - 1));

    f32x16 acc[4][NT];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[c][nt][i] = 0.f;

    // (class, tap, input offset) table: out(2i+py,2j+px) <- in(i+di,j+dj) * W[tap]
    constexpr int NTAP = 9;
    constexpr int tap_cls[NTAP] = {0, 1, 1, 2, 2, 3, 3, 3, 3};
    constexpr int tap_t[NTAP] = {4, 5, 3, 7, 1, 8, 6, 2, 0};
    constexpr int tap_off[NTAP] = {0, 0, 1, 0, 2, 0, 1, 2, 3};  // di*2+dj

    for (int c0 = 0; c0 < Cin; c0 += CK) {
        __syncthreads();
        stage_items<T, 3>(tid, npix * 4, a.src0, a.src1, TWO_SRC, cf, Cin, a.slope, patch,
                          [&](int it, bool& ok, size_t& g, int& loff, int& cb) {
            const int pix = it >> 2, q = it & 3;
            const int img = fastdiv(pix, a.m_pp), rem = pix - img * PP, py = fastdiv(rem, a.m_pw), px = rem - py * PW;
            const int b = b0 + img, iy = iy0 + py, ix = ix0 + px;
            loff = pix * PATCH_PITCH + q * 16; cb = c0 + q * E16;
            ok = b < a.B && iy < Hs && ix < Ws;
            g = (((size_t)b * Hs + iy) * Ws + ix) * Cin + cb;
        });
        __syncthreads();
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            Frag<T> af[4];
#pragma unroll
            for (int o = 0; o < 4; ++o)
                af[o] = load_frag(reinterpret_cast<const T*>(patch + (pbase + (o >> 1) * PW + (o & 1)) * PATCH_PITCH + ks * 32) + h * 8);
#pragma unroll
            for (int k = 0; k < NTAP; ++k) {
                const size_t kg = (size_t)tap_t[k] * (Cin >> 3) + ((c0 + ks * 16) >> 3) + h;
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    Frag<T> bf = load_frag(a.wp + (kg * Cout + n0 + nt * 32 + r) * 8);
                    mma(acc[tap_cls[k]][nt], af[tap_off[k]], bf);
                }
            }
        }
    }

#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int n = n0 + nt * 32 + r;
        float bv = 0.f, sc = 0.f, sh = 0.f, is = 0.f, xm = 0.f, s1 = 0.f, s2 = 0.f;
        if (EPI == EPI_FWD) bv = a.bias ? a.bias[n] : 0.f;
        if (EPI == EPI_BWD) {
            sc = a.ocoef[LC_SC * Cout + n]; sh = a.ocoef[LC_SH * Cout + n];
            is = a.ocoef[LC_INVSTD * Cout + n]; xm = a.ocoef[LC_XM * Cout + n];
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int RR = wave * 32 + acc_row(i, lane);
                const int b = b0 + (RR >> (a.lth + a.ltw));
                if (b < a.B) {
                    const int oy = 2 * (iy0 + ((RR >> a.ltw) & (th - 1))) + (c >> 1);
                    const int ox = 2 * (ix0 + (RR & (tw - 1))) + (c & 1);
                    const size_t idx = (((size_t)b * 2 * Hs + oy) * 2 * Ws + ox) * Cout + n;
                    epi_store<T>(a, idx, n, acc[c][nt][i], bv, sc, sh, is, xm, s1, s2);
                }
            }
        }
        if (EPI != EPI_PLAIN) {
            s1 += __shfl_xor(s1, 32, 64); s2 += __shfl_xor(s2, 32, 64);
            if (h == 0) { red[((wave * NT + nt) * 32 + r) * 2] = s1; red[((wave * NT + nt) * 32 + r) * 2 + 1] = s2; }
        }
    }
    if (EPI != EPI_PLAIN) {
        __syncthreads();
        if (tid < NT * 32) {
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int w = 0; w < 4; ++w) { s1 += red[((w * NT) * 32 + tid) * 2]; s2 += red[((w * NT) * 32 + tid) * 2 + 1]; }
            double* st_ = a.stat + stat_rep() * 2 * Cout;
            unsafeAtomicAdd(&st_[n0 + tid], (double)s1);
            unsafeAtomicAdd(&st_[Cout + n0 + tid], (double)s2);
        }
    }
}

// ---------------------------------------------------------------------------
// Weight gradient.  K of the GEMM = low-res pixels; both operands are staged in
// LDS as [pixel][channel] and read k-major: bf16 through ds_read_b64_tr_b16 (the
// hardware transpose read), f32 as one dword per k (the 32x32x2 f32 MFMA takes a
// single k per lane, so no transpose is needed).
template <typename T> struct WgradArgs {
    const T* s0; const T* s1; const float* scoef; float sslope;  // low-res operand  [B,Hs,Ws,CA]
    const T* g0; const T* g1; const float* gcoef; float gslope;  // high-res operand [B,2Hs,2Ws,CB]
    float* slab;                                                 // [nsplit][9][CA][CB]
    int s_two, g_two;
    int B, Hs, Ws, CA, CB;
    int lth, ltw, lTB, tiles_x, tiles_y, n_tiles, tiles_per_split;
    int use_tr16, rev;
    unsigned m_pp, m_pw, m_tx, m_txy;
    BnFuse fuse;                                                 // mode == BNF_BWD: the gradient operand's coefficients come from batch statistics
    long long* dbg;                                              // diagnostic builds only (wgrad_split_kernel): per-wave cycle counters
};


typedef __attribute__((ext_vector_type(8))) short s16x8;
// k-major 16-bit fragment from two transposed LDS reads (rows k..k+3 and k+4..k+7 of a [k][channel] image)
template <typename T>
__device__ __forceinline__ Frag<T> frag_tr16(const char* ad0, const char* ad1) {
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_PTR(s16x4))ad0);
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_PTR(s16x4))ad1);
    const s16x8 w = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    Frag<T> f;
    f.v = __builtin_bit_cast(typename H16<T>::v8, w);
    return f;
}

// PRE = true: the next K tile's operand chunks are loaded into registers while the current tile is in
// the matrix pipe (persistent loop over this workgroup's K tiles).  CONVT selects which side carries
// the two-source gradient operand (ConvTranspose2d: high-res side; Conv2d: low-res side).
// NW = waves per workgroup (4, or 8 for the 128x32-channel tile of the wide layers: twice the channels per staged byte).
// RAW (prefetching variants): both operands are materialised (already transformed and rounded): staged as plain copies.
template <typename T, int WA, int WB, bool CONVT, bool PRE, int NW = 4, bool RAW = false>
__global__ __launch_bounds__(64 * NW, ((sizeof(T) == 2 && WA * WB < 4 && NW == 4) ? 2 : 1)) void wgrad_kernel(WgradArgs<T> a) {
    static_assert(!RAW || PRE, "RAW operands: prefetching variants only");
    // NW waves = WA x WB channel blocks x TS tap groups; a wave owns taps ts, ts+TS, ... (no cross-wave sum)
    static_assert(NW == 4 || PRE, "the 8-wave layout exists for the prefetching variant only");
    constexpr int NTHR = 64 * NW, SIT = WG_KP * 4 * WA / NTHR;   // threads; low-res chunks per thread
    constexpr int TS = NW / (WA * WB), NTW = (9 + TS - 1) / TS, E16 = 16 / sizeof(T);
    constexpr int SROW = 32 * WA * sizeof(T), GROW = 32 * WB * sizeof(T);
    // row pitches of the two staged operands: the transposed reads of a lane group touch 4 pixel rows x 32..64 B - consecutive rows
    // of the low-res tile, every second row of the high-res patch - and must land on distinct banks (pitch = 16 dwords mod 64 for
    // the former, 2 * pitch = 16 or 48 dwords mod 64 for the latter); with the 16-byte pads of round 2 half of the LDS cycles of this
    // kernel were bank conflicts (profiles/r02_pmc_sq_v2.txt)
    constexpr int SPITCH = SROW + WG_SPAD, GPITCH = GROW + WG_GPAD;
    constexpr int SCH = SROW / 16, GCH = GROW / 16;  // 16-byte chunks per staged pixel
    constexpr bool S_TWO = !CONVT && !RAW, G_TWO = CONVT && !RAW;
    constexpr int MAXG = (5 * WB * 256 + NTHR - 1) / NTHR;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, h = lane >> 5;
    const int th = 1 << a.lth, tw = 1 << a.ltw, TB = 1 << a.lTB;
    const int PH = 2 * th + 1, PW = 2 * tw + 1, PP = PH * PW, npix = TB * PP;
    const int Hs = a.Hs, Ws = a.Ws, Hg = 2 * a.Hs, Wg = 2 * a.Ws, CA = a.CA, CB = a.CB;
    const int a0 = blockIdx.y * 32 * WA, bc0 = blockIdx.z * 32 * WB;
    const int wa = wave % WA, wb = (wave / WA) % WB, ts = wave / (WA * WB);

    float* cfs = reinterpret_cast<float*>(smem);             // [3][32*WA]
    float* cfg = cfs + 3 * 32 * WA;                          // [3][32*WB]
    char* stile = reinterpret_cast<char*>(cfg + 3 * 32 * WB);  // [64][SPITCH]
    char* gtile = stile + WG_KP * SPITCH;                    // [npix][GPITCH]
    // tile-independent staging table of the high-res patch: {relative element offset, LDS offset/16 | top<<13 | left<<14 | img<<15}
    int2* gtab = reinterpret_cast<int2*>(gtile + npix * GPITCH);
    if constexpr (PRE) {   // (padded to MAXG*256 entries; padding entries carry image 0xffff, which never passes the batch test)
        for (int it = tid; it < max(npix * GCH, MAXG * NTHR); it += NTHR) {
            const int pix = it / GCH, qq = it - pix * GCH;
            const int img = fastdiv(pix, a.m_pp), rem = pix - img * PP, py = fastdiv(rem, a.m_pw), px = rem - py * PW;
            gtab[it] = it < npix * GCH ? make_int2(((img * Hg + py) * Wg + px) * CB + qq * E16,
                                                   ((pix * GPITCH + qq * 16) >> 4) | ((py == 0) << 13) | ((px == 0) << 14) | (img << 15))
                                       : make_int2(0, 0xffff << 15);
        }
    }

    // staging coefficients (tile-local rows); the gradient operand's may be derived here from the batch statistics
    if constexpr (RAW) {   // identity (the synchronous remainder path still goes through the transform helper)
        for (int i = tid; i < 3 * 32 * WA; i += NTHR) cfs[i] = i < 32 * WA ? 1.f : 0.f;
        for (int i = tid; i < 3 * 32 * WB; i += NTHR) cfg[i] = i < 32 * WB ? 1.f : 0.f;
    } else {
    if (a.fuse.mode == BNF_BWD && a.s_two) {
        for (int i = tid; i < 32 * WA; i += NTHR) bn_fused_channel(a.fuse, a0 + i, false, cfs[i], cfs[32 * WA + i], cfs[2 * 32 * WA + i]);
    } else {
        for (int i = tid; i < 3 * 32 * WA; i += NTHR) cfs[i] = a.scoef[(i / (32 * WA)) * CA + a0 + (i % (32 * WA))];
    }
    if (a.fuse.mode == BNF_BWD && a.g_two) {
        for (int i = tid; i < 32 * WB; i += NTHR) bn_fused_channel(a.fuse, bc0 + i, false, cfg[i], cfg[32 * WB + i], cfg[2 * 32 * WB + i]);
    } else {
        for (int i = tid; i < 3 * 32 * WB; i += NTHR) cfg[i] = a.gcoef[(i / (32 * WB)) * CB + bc0 + (i % (32 * WB))];
    }
    }

    f32x16 acc[NTW];
#pragma unroll
    for (int t = 0; t < NTW; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;

    // lane geometry for k-major reads
    const int g4 = lane >> 4, li = lane & 15, q = li >> 2, p = li & 3;

    const int t_begin = blockIdx.x * a.tiles_per_split;
    const int t_end = min(a.n_tiles, t_begin + a.tiles_per_split);

    auto s_map = [&](int b0, int y0, int x0, int it, bool& ok, size_t& g, int& loff, int& cb) {
        const int k = it / SCH, qq = it - k * SCH;
        const int b = b0 + (k >> (a.lth + a.ltw)), y = y0 + ((k >> a.ltw) & (th - 1)), x = x0 + (k & (tw - 1));
        loff = k * SPITCH + qq * 16; cb = qq * E16; ok = b < a.B;
        g = (((size_t)b * Hs + y) * Ws + x) * CA + a0 + qq * E16;
    };
    auto g_map = [&](int b0, int y0, int x0, int it, bool& ok, size_t& g, int& loff, int& cb) {
        const int pix = it / GCH, qq = it - pix * GCH;
        const int img = fastdiv(pix, a.m_pp), rem = pix - img * PP, py = fastdiv(rem, a.m_pw), px = rem - py * PW;
        const int b = b0 + img, iy = 2 * y0 - 1 + py, ix = 2 * x0 - 1 + px;
        loff = pix * GPITCH + qq * 16; cb = qq * E16;
        ok = b < a.B && iy >= 0 && iy < Hg && ix >= 0 && ix < Wg;
        g = (((size_t)b * Hg + iy) * Wg + ix) * CB + bc0 + qq * E16;
    };
    auto tile_origin = [&](int tile_, int& b0, int& y0, int& x0) {
        const int tile = a.rev ? a.n_tiles - 1 - tile_ : tile_;   // reversed walk (see ConvArgs::rev)
        const int bt = fastdiv(tile, a.m_txy), trem = tile - bt * a.tiles_x * a.tiles_y, ty = fastdiv(trem, a.m_tx), tx = trem - ty * a.tiles_x;
        b0 = bt << a.lTB; y0 = ty << a.lth; x0 = tx << a.ltw;
    };
    // prefetch registers.  A thread always stages the same 16-byte channel quarter of both operands (256 is a
    // multiple of the chunks per pixel), so its coefficients live in registers; offsets are 32-bit bytes.
    Vec16<T> ps0[PRE ? SIT : 1], ps1[(PRE && S_TWO) ? SIT : 1], pg0[PRE ? MAXG : 1], pg1[(PRE && G_TWO) ? MAXG : 1];
    int srel[PRE ? SIT : 1], sloff[PRE ? SIT : 1];     // low-res operand: tile-independent element offset / LDS offset | image << 20
    int gmeta[PRE ? MAXG : 1];                        // high-res operand: table word of the chunk (LDS offset, halo flags, image)
    constexpr int NE = Vec16<T>::N;
    f32x2 ks0[PRE ? NE / 2 : 1], ks1[(PRE && S_TWO) ? NE / 2 : 1], ks2[PRE ? NE / 2 : 1];
    f32x2 kg0[PRE ? NE / 2 : 1], kg1[(PRE && G_TWO) ? NE / 2 : 1], kg2[PRE ? NE / 2 : 1];
    if constexpr (PRE) {
        __syncthreads();   // coefficient rows / table published
#pragma unroll
        for (int u = 0; u < SIT; ++u) {
            const int it = tid + u * NTHR, k = it / SCH, qq = it - k * SCH;
            const int img = k >> (a.lth + a.ltw), y = (k >> a.ltw) & (th - 1), x = k & (tw - 1);
            srel[u] = ((img * Hs + y) * Ws + x) * CA + a0 + qq * E16;
            sloff[u] = (k * SPITCH + qq * 16) | (img << 20);
        }
        const int sq = (tid % SCH) * E16, gq = (tid % GCH) * E16;
#pragma unroll
        for (int e = 0; e < NE / 2; ++e) {
            ks0[e] = f32x2{cfs[sq + 2 * e], cfs[sq + 2 * e + 1]}; ks2[e] = f32x2{cfs[2 * 32 * WA + sq + 2 * e], cfs[2 * 32 * WA + sq + 2 * e + 1]};
            if constexpr (S_TWO) ks1[e] = f32x2{cfs[32 * WA + sq + 2 * e], cfs[32 * WA + sq + 2 * e + 1]};
            kg0[e] = f32x2{cfg[gq + 2 * e], cfg[gq + 2 * e + 1]}; kg2[e] = f32x2{cfg[2 * 32 * WB + gq + 2 * e], cfg[2 * 32 * WB + gq + 2 * e + 1]};
            if constexpr (G_TWO) kg1[e] = f32x2{cfg[32 * WB + gq + 2 * e], cfg[32 * WB + gq + 2 * e + 1]};
        }
    }
    // v0*k0 (+ v1*k1) + k2, LeakyReLU on the activation operand; pairs -> packed f32 math
    auto xform2 = [&](const Vec16<T>& v0, const Vec16<T>& v1, const f32x2* k0, const f32x2* k1, const f32x2* k2, bool two, float slope)
        __attribute__((always_inline)) {
        if constexpr (RAW) return v0;
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
        for (int u = 0; u < (PRE ? SIT : 0); ++u) {
            const uint32_t g = (b0 + (sloff[u] >> 20)) < a.B ? (uint32_t)(sbase + srel[u]) * (uint32_t)sizeof(T) : 0u;
            ps0[u] = *reinterpret_cast<const Vec16<T>*>(reinterpret_cast<const char*>(a.s0) + g);
            if constexpr (S_TWO) ps1[u] = *reinterpret_cast<const Vec16<T>*>(reinterpret_cast<const char*>(a.s1) + g);
        }
        const int gbase = ((b0 * Hg + 2 * y0 - 1) * Wg + 2 * x0 - 1) * CB + bc0;
        const int tmask = (y0 == 0 ? 1 << 13 : 0) | (x0 == 0 ? 1 << 14 : 0), nb = a.B - b0;   // uniform per tile
        int2 e[PRE ? MAXG : 1];
#pragma unroll
        for (int u = 0; u < (PRE ? MAXG : 0); ++u) e[u] = gtab[tid + u * NTHR];
#pragma unroll
        for (int u = 0; u < (PRE ? MAXG : 0); ++u) {
            const bool ok = ((e[u].y & tmask) == 0) & ((e[u].y >> 15) < nb);
            gmeta[u] = ok ? e[u].y : (e[u].y | (1 << 31));
            const uint32_t g = ok ? (uint32_t)(gbase + e[u].x) * (uint32_t)sizeof(T) : 0u;
            pg0[u] = *reinterpret_cast<const Vec16<T>*>(reinterpret_cast<const char*>(a.g0) + g);
            if constexpr (G_TWO) pg1[u] = *reinterpret_cast<const Vec16<T>*>(reinterpret_cast<const char*>(a.g1) + g);
        }
    };
    if constexpr (PRE) { if (t_begin < t_end) issue_tile(t_begin); }
    for (int tile = t_begin; tile < t_end; ++tile) {
        int b0, y0, x0; tile_origin(tile, b0, y0, x0);
        if constexpr (!PRE) {
            __syncthreads();
            // coefficient rows are stored tile-local (stride 32*WA / 32*WB), channel index = chunk*E16
            stage_items<T, WA>(tid, WG_KP * SCH, a.s0, a.s1, S_TWO, cfs, 32 * WA, a.sslope, stile,
                              [&](int it, bool& ok, size_t& g, int& loff, int& cb) { s_map(b0, y0, x0, it, ok, g, loff, cb); });
            stage_items<T, 4>(tid, npix * GCH, a.g0, a.g1, G_TWO, cfg, 32 * WB, a.gslope, gtile,
                              [&](int it, bool& ok, size_t& g, int& loff, int& cb) { g_map(b0, y0, x0, it, ok, g, loff, cb); });
            __syncthreads();
        } else {
            __syncthreads();                       // previous tile consumed
            // transform + store this tile's chunks and, as each register pair becomes free, request the same chunk of the
            // next tile (the loads then fly during the rest of the staging phase as well as the MFMA phase)
            const bool nh = tile + 1 < t_end;
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
            int2 e[MAXG];
#pragma unroll
            for (int u = 0; u < MAXG; ++u) e[u] = gtab[tid + u * NTHR];
#pragma unroll
            for (int u = 0; u < MAXG; ++u) {
                Vec16<T> o = xform2(pg0[u], pg1[G_TWO ? u : 0], kg0, kg1, kg2, G_TWO, a.gslope);
                if (gmeta[u] < 0) o = zero_vec16<T>();
                if (tid + u * NTHR < npix * GCH) *reinterpret_cast<Vec16<T>*>(gtile + ((gmeta[u] & 0x1fff) << 4)) = o;
                const bool ok = nh & ((e[u].y & ntmask) == 0) & ((e[u].y >> 15) < nnb);
                gmeta[u] = ok ? e[u].y : (e[u].y | (1 << 31));
                const uint32_t g = ok ? (uint32_t)(ngbase + e[u].x) * (uint32_t)sizeof(T) : 0u;
                pg0[u] = *reinterpret_cast<const Vec16<T>*>(reinterpret_cast<const char*>(a.g0) + g);
                if constexpr (G_TWO) pg1[u] = *reinterpret_cast<const Vec16<T>*>(reinterpret_cast<const char*>(a.g1) + g);
            }
            // patches with more than MAXG*256 chunks (many small images per tile): synchronous remainder
            for (int it = tid + MAXG * NTHR; it < npix * GCH; it += NTHR) {
                bool ok; size_t g; int loff, cb;
                g_map(b0, y0, x0, it, ok, g, loff, cb);
                Vec16<T> v = zero_vec16<T>();
                if (ok) v = load_transform16<T>(a.g0, a.g1, G_TWO, g, cfg, 32 * WB, cb, a.gslope);
                *reinterpret_cast<Vec16<T>*>(gtile + loff) = v;
            }
            __syncthreads();                       // tile published
        }

#pragma unroll 1
        for (int ks = 0; ks < WG_KP / 16; ++ks) {
            Frag<T> af;
            int growbase[8];  // patch pixel index (tap 0,0) of the k rows this lane addresses
            if constexpr (sizeof(T) == 4) {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int k = ks * 16 + 8 * h + j;
                    af.v[j] = *reinterpret_cast<const float*>(stile + k * SPITCH + (wa * 32 + r) * 4);
                    growbase[j] = ((k >> (a.lth + a.ltw)) * PH + 2 * ((k >> a.ltw) & (th - 1))) * PW + 2 * (k & (tw - 1));
                }
            } else {
                if (a.use_tr16) {
                    const int k0 = ks * 16 + 8 * (g4 >> 1) + q, k1 = k0 + 4;
                    const int col = (wa * 32 + 16 * (g4 & 1) + 4 * p) * 2;
                    af = frag_tr16<T>(stile + k0 * SPITCH + col, stile + k1 * SPITCH + col);
                    growbase[0] = ((k0 >> (a.lth + a.ltw)) * PH + 2 * ((k0 >> a.ltw) & (th - 1))) * PW + 2 * (k0 & (tw - 1));
                    growbase[1] = ((k1 >> (a.lth + a.ltw)) * PH + 2 * ((k1 >> a.ltw) & (th - 1))) * PW + 2 * (k1 & (tw - 1));
                } else {
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const int k = ks * 16 + 8 * h + j;
                        af.v[j] = *reinterpret_cast<const T*>(stile + k * SPITCH + (wa * 32 + r) * 2);
                        growbase[j] = ((k >> (a.lth + a.ltw)) * PH + 2 * ((k >> a.ltw) & (th - 1))) * PW + 2 * (k & (tw - 1));
                    }
                }
            }
#pragma unroll
            for (int ti = 0; ti < NTW; ++ti) {
                const int t = ts + ti * TS;
                if (t < 9) {   // wave-uniform
                    const int ky = (t * 11) >> 5, kx = t - 3 * ky;
                    const int toff = ky * PW + kx;
                    Frag<T> bf;
                    if constexpr (sizeof(T) == 4) {
#pragma unroll
                        for (int j = 0; j < 8; ++j)
                            bf.v[j] = *reinterpret_cast<const float*>(gtile + (growbase[j] + toff) * GPITCH + (wb * 32 + r) * 4);
                    } else {
                        if (a.use_tr16) {
                            const int col = (wb * 32 + 16 * (g4 & 1) + 4 * p) * 2;
                            bf = frag_tr16<T>(gtile + (growbase[0] + toff) * GPITCH + col, gtile + (growbase[1] + toff) * GPITCH + col);
                        } else {
#pragma unroll
                            for (int j = 0; j < 8; ++j)
                                bf.v[j] = *reinterpret_cast<const T*>(gtile + (growbase[j] + toff) * GPITCH + (wb * 32 + r) * 2);
                        }
                    }
                    mma(acc[ti], af, bf);
                }
            }
        }
    }

    // partial slab: rows = low-res-side channel (a), lanes = high-res-side channel (b)
    const size_t slab_id = blockIdx.x;
#pragma unroll
    for (int ti = 0; ti < NTW; ++ti) {
        const int t = ts + ti * TS;
        if (t < 9) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int ca = a0 + wa * 32 + acc_row(i, lane), cb = bc0 + wb * 32 + r;
                a.slab[((slab_id * 9 + t) * CA + ca) * CB + cb] = acc[ti][i];
            }
        }
    }
}

// out[(a*CB+b)*9+t] = sum_s slab[s][t][a][b]   (CA>0: conv weight layout [A][B][3][3])
// out[j]            = sum_s slab[s][j]          (CA==0)
// block = 64 outputs x 4 slab groups; each thread keeps 8 independent loads in flight.
// blockIdx.y = slab range [y*per, (y+1)*per) (per = nslab for a single-level reduction); a ranged launch writes its partial
// sums to out + y*n (CA = 0 layout)
static __global__ __launch_bounds__(256) void reduce_slab_kernel(const float* __restrict__ slab, int nslab_all, int n,
                                                                 float* __restrict__ out, int CA, int CB, float scale, int per) {
    __shared__ float part[4][64];
    const int jl = threadIdx.x & 63, g = threadIdx.x >> 6, j = blockIdx.x * 64 + jl;
    slab += (size_t)blockIdx.y * per * n; out += (size_t)blockIdx.y * n;
    const int nslab = min(per, nslab_all - (int)blockIdx.y * per);
    float s = 0.f;
    if (j < n) {
        int k = g;
        for (; k + 28 < nslab; k += 32) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = slab[(size_t)(k + 4 * u) * n + j];
#pragma unroll
            for (int u = 0; u < 8; ++u) s += v[u];
        }
        for (; k < nslab; k += 4) s += slab[(size_t)k * n + j];
    }
    part[g][jl] = s;
    __syncthreads();
    if (g == 0 && j < n) {
        s = (part[0][jl] + part[1][jl] + part[2][jl] + part[3][jl]) * scale;
        if (CA > 0) {
            const int t = j / (CA * CB), rem = j - t * CA * CB;
            out[(size_t)rem * 9 + t] = s;
        } else {
            out[j] = s;
        }
    }
}


// ---------------------------------------------------------------------------
// Output conv forward + sigmoid + BCE + dlogit on MFMA (bf16 mode); same math as convout_fwd_kernel.
//   part[p][t] = sum_c a[p][c] w[t][c]   : M = patch pixels (tile + 1-pixel halo), K = 32 channels, N = 9 taps
//   logit[q]   = bias + sum_t part[q + off(t)][t]
// Persistent workgroups over 8x32-pixel tiles; next tile's (8+2)x(32+2) patch of y is prefetched.
template <typename T> struct ConvOutFwdMfmaArgs {
    const T* yf; const float* coef; const float* wt; const float* bias; const float* target;
    float* xhat; float* dlogit; double* accum;
    int B, H, W, n_tiles; float inv_n, slope;
    BnFuse fuse;
    int rev;
};

template <typename T>
__global__ __launch_bounds__(256, 2) void convout_fwd_mfma_kernel(ConvOutFwdMfmaArgs<T> a) {
    typedef typename H16<T>::v8 T8;
    constexpr int TH = 8, TW = 32, PH = TH + 2, PW = TW + 2, NP = PH * PW, NPAD = 384, PITCH = 80, NCHK = NP * 4, MAXI = 6;
    __shared__ __attribute__((aligned(16))) char atile[NPAD * PITCH];
    __shared__ float part[NPAD * 9];
    __shared__ float cf[64];
    __shared__ float wred[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, h = lane >> 5;
    const int tiles_x = a.W / TW, tiles_y = a.H / TH;
    if (tid < 32) {
        if (a.fuse.mode == BNF_FWD) { float k1; bn_fused_channel(a.fuse, tid, blockIdx.x == 0, cf[tid], k1, cf[32 + tid]); }
        else { cf[tid] = a.coef[tid]; cf[32 + tid] = a.coef[64 + tid]; }
    }
    // rows NP..NPAD of the patch image stay zero
    for (int i = tid; i < (NPAD - NP) * PITCH / 16; i += 256) *reinterpret_cast<f32x4*>(atile + NP * PITCH + i * 16) = f32x4{0.f, 0.f, 0.f, 0.f};
    // B operand: B[k = channel][col = tap r]
    Frag<T> wf[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int j = 0; j < 8; ++j) wf[ks].v[j] = (T)(r < 9 ? a.wt[r * 32 + ks * 16 + 8 * h + j] : 0.f);
    const float bo = a.bias[0];
    float bsum = 0.f;

    auto tile_origin = [&](int tile_, int& b, int& y0, int& x0) {
        const int tile = a.rev ? a.n_tiles - 1 - tile_ : tile_;   // reversed walk: start with what the producer wrote last
        const int tx = tile % tiles_x, ty = (tile / tiles_x) % tiles_y;
        b = tile / (tiles_x * tiles_y); y0 = ty * TH; x0 = tx * TW;
    };
    T8 pre[MAXI]; int ok[MAXI]; float pretg;
    // chunk id = tid + 256u: patch pixel id>>2, channel quarter id&3 = tid&3 - the same quarter for every chunk of a thread
    auto prefetch = [&](int tile) {
        int b, y0, x0; tile_origin(tile, b, y0, x0);
        pretg = a.target[((size_t)b * a.H + y0 + (tid >> 5)) * a.W + x0 + (tid & 31)];
        const int base = ((b * a.H + y0 - 1) * a.W + x0 - 1) * 32 + (tid & 3) * 8;
#pragma unroll
        for (int u = 0; u < MAXI; ++u) {
            const int id = tid + 256 * u, pix = id >> 2;
            const int py = pix / PW, px = pix - py * PW, gy = y0 - 1 + py, gx = x0 - 1 + px;
            ok[u] = id < NCHK && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
            const uint32_t g = ok[u] ? (uint32_t)(base + (py * a.W + px) * 32) * 2u : 0u;   // byte offset (< 4 GiB: host check)
            pre[u] = *reinterpret_cast<const T8*>(reinterpret_cast<const char*>(a.yf) + g);
        }
    };
    __syncthreads();                     // cf published
    f32x2 kc[4], kh[4];                  // this thread's 8 channels: scale / shift pairs
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        kc[e] = f32x2{cf[(tid & 3) * 8 + 2 * e], cf[(tid & 3) * 8 + 2 * e + 1]};
        kh[e] = f32x2{cf[32 + (tid & 3) * 8 + 2 * e], cf[32 + (tid & 3) * 8 + 2 * e + 1]};
    }

    int tile = blockIdx.x;
    if (tile < a.n_tiles) prefetch(tile);
    for (; tile < a.n_tiles; tile += gridDim.x) {
        int b, y0, x0; tile_origin(tile, b, y0, x0);
        __syncthreads();
#pragma unroll
        for (int u = 0; u < MAXI; ++u) {
            const int id = tid + 256 * u;
            T8 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                f32x2 z = f32x2{(float)pre[u][2 * e], (float)pre[u][2 * e + 1]} * kc[e] + kh[e];
                const f32x2 zs = z * a.slope;
                z.x = fmaxf(z.x, zs.x); z.y = fmaxf(z.y, zs.y);
                o[2 * e] = (T)z.x; o[2 * e + 1] = (T)z.y;
            }
            if (!ok[u]) o = T8{0, 0, 0, 0, 0, 0, 0, 0};
            if (id < NCHK) *reinterpret_cast<T8*>(atile + (id >> 2) * PITCH + (id & 3) * 16) = o;
        }
        __syncthreads();
        const float tg = pretg;
        if (tile + (int)gridDim.x < a.n_tiles) prefetch(tile + gridDim.x);
        // 12 row blocks of 32 patch pixels, 3 per wave
#pragma unroll
        for (int mb = 0; mb < 3; ++mb) {
            const int row0 = (wave * 3 + mb) * 32;
            f32x16 acc;
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                Frag<T> af = load_frag(reinterpret_cast<const T*>(atile + (row0 + r) * PITCH + ks * 32) + h * 8);
                mma(acc, af, wf[ks]);
            }
            if (r < 9) {
#pragma unroll
                for (int i = 0; i < 16; ++i) part[(row0 + acc_row(i, lane)) * 9 + r] = acc[i];
            }
        }
        __syncthreads();
        // 256 output pixels, one per thread
        {
            const int oy = tid >> 5, ox = tid & 31;
            float logit = bo;
#pragma unroll
            for (int t = 0; t < 9; ++t) logit += part[((oy + t / 3) * PW + ox + t % 3) * 9 + t];
            const size_t gi = ((size_t)b * a.H + y0 + oy) * a.W + x0 + ox;
            const float xh = 1.f / (1.f + expf(-logit));
            const float l1 = fmaxf(logf(xh), -100.f), l0 = fmaxf(logf(1.f - xh), -100.f);
            bsum += -(tg * l1 + (1.f - tg) * l0);
            const float om = xh * (1.f - xh);
            a.xhat[gi] = xh;
            a.dlogit[gi] = (xh - tg) / fmaxf(om, 1e-12f) * om * a.inv_n;
        }
    }
    bsum = wave_sum(bsum);
    if (lane == 0) wred[wave] = bsum;
    __syncthreads();
    if (tid == 0) unsafeAtomicAdd(&a.accum[stat_rep() * 8 + 0], (double)(wred[0] + wred[1] + wred[2] + wred[3]));
}

// ---------------------------------------------------------------------------
// Output-conv backward on MFMA (bf16 mode).  Same math as convout_bwd_kernel (edge_kernels.cuh):
//   dA[p][c] = sum_t dl[p-off(t)] w[t][c]        -> one 32x32x16 MFMA per 32 pixels (K = 9 taps, padded)
//   dW[c][t] += sum_p a[p][c] dl[p-off(t)]       -> K = pixels, A operand read k-major (tr16) from the y tile
//   dz = dA * leaky'(z), per-channel sum dz, sum dz*xhat, sum dl
// Persistent workgroups walk 8x32-pixel tiles; the next tile's y is prefetched into registers while the
// current one is in the matrix pipe; dz goes out through LDS as whole 64-byte pixels.
template <typename T> struct ConvOutBwdMfmaArgs {
    const T* yf; const float* ocoef; const float* wt; const float* dlogit; const float* gscale;
    T* dz; float* slab; double* stat; double* dbias;
    int B, H, W, n_tiles; float slope;
    int rev; float gmul;
    int store_dz;   // 0: statistics / weight gradient only (the consumer recomputes dz from dlogit: conv_fused.cuh, RECOMP)
};

template <typename T>
__global__ __launch_bounds__(256, 2) void convout_bwd_mfma_kernel(ConvOutBwdMfmaArgs<T> a) {
    typedef typename H16<T>::v8 T8;
    constexpr int TH = 8, TW = 32, PITCH = 80, DW = TW + 2, DH = TH + 2;
    __shared__ __attribute__((aligned(16))) char ytile[TH * TW * PITCH];
    __shared__ float dl_s[DH * DW + 6];
    __shared__ float red[4][32 * 9 + 64 + 1];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, h = lane >> 5;
    const int g4 = lane >> 4, li = lane & 15, q = li >> 2, p = li & 3;
    const int tiles_x = a.W / TW, tiles_y = a.H / TH;
    const float gs = (a.gscale ? a.gscale[0] : 1.f) * a.gmul;
    const float sc = a.ocoef[LC_SC * 32 + r], sh = a.ocoef[LC_SH * 32 + r];
    const float is = a.ocoef[LC_INVSTD * 32 + r], xm = a.ocoef[LC_XM * 32 + r];

    // weights as the B operand of dA: B[k = tap][col = channel r]
    Frag<T> wfrag;
#pragma unroll
    for (int j = 0; j < 8; ++j) { const int t = 8 * h + j; wfrag.v[j] = (T)(t < 9 ? a.wt[t * 32 + r] : 0.f); }

    f32x16 accw;
#pragma unroll
    for (int i = 0; i < 16; ++i) accw[i] = 0.f;
    float s1 = 0.f, s2 = 0.f, sdl = 0.f;

    auto tile_origin = [&](int tile_, int& b, int& y0, int& x0) {
        const int tile = a.rev ? a.n_tiles - 1 - tile_ : tile_;   // reversed walk: start with what the producer wrote last
        const int tx = tile % tiles_x, ty = (tile / tiles_x) % tiles_y;
        b = tile / (tiles_x * tiles_y); y0 = ty * TH; x0 = tx * TW;
    };
    // each thread owns 4 of the tile's 1024 16-byte chunks: chunk id = tid + 256*u -> pixel id>>2, quarter id&3
    T8 pre[4];
    float predl[2];     // and up to 2 of the (TH+2)x(TW+2) dlogit values
    auto prefetch = [&](int tile) {
        int b, y0, x0; tile_origin(tile, b, y0, x0);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int id = tid + 256 * u, pix = id >> 2, qq = id & 3;
            const size_t g = (((size_t)b * a.H + y0 + (pix >> 5)) * a.W + x0 + (pix & 31)) * 32 + qq * 8;
            pre[u] = *reinterpret_cast<const T8*>(a.yf + g);
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int i = tid + 256 * u, rr = i / DW, cc = i - rr * DW, gy = y0 - 1 + rr, gx = x0 - 1 + cc;
            const bool in = i < DH * DW && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
            const float v = a.dlogit[in ? ((size_t)b * a.H + gy) * a.W + gx : 0];
            predl[u] = in ? v * gs : 0.f;
        }
    };

    int tile = blockIdx.x;
    if (tile < a.n_tiles) prefetch(tile);
    for (; tile < a.n_tiles; tile += gridDim.x) {
        int b, y0, x0; tile_origin(tile, b, y0, x0);
        __syncthreads();   // previous tile fully consumed
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int id = tid + 256 * u;
            *reinterpret_cast<T8*>(ytile + (id >> 2) * PITCH + (id & 3) * 16) = pre[u];
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int i = tid + 256 * u, rr = i / DW, cc = i - rr * DW;
            if (i < DH * DW) {
                dl_s[i] = predl[u];
                if (rr >= 1 && rr <= TH && cc >= 1 && cc <= TW) sdl += predl[u];
            }
        }
        __syncthreads();
        if (tile + (int)gridDim.x < a.n_tiles) prefetch(tile + gridDim.x);   // in flight during the MFMAs below

        // ---- dA: 2 blocks of 32 pixels per wave (tile rows 2*wave, 2*wave+1)
        f32x16 acca[2];
#pragma unroll
        for (int mb = 0; mb < 2; ++mb) {
#pragma unroll
            for (int i = 0; i < 16; ++i) acca[mb][i] = 0.f;
            const int ly = 2 * wave + mb, lx = r;
            Frag<T> af;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int t = 8 * h + j, tt = t < 9 ? t : 0;
                const float v = dl_s[(ly - tt / 3 + 2) * DW + (lx - tt % 3 + 2)];
                af.v[j] = (T)(t < 9 ? v : 0.f);
            }
            mma(acca[mb], af, wfrag);
        }
        // ---- dW: K = the wave's 64 pixels, A = a^T (k-major via tr16 + BN/LeakyReLU), B = dl taps
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int k0 = wave * 64 + ks * 16 + 8 * (g4 >> 1) + q, k1 = k0 + 4;
            const int col = (16 * (g4 & 1) + 4 * p) * 2;
            Frag<T> yfrag = frag_tr16<T>(ytile + k0 * PITCH + col, ytile + k1 * PITCH + col);
            Frag<T> afr, bfr;
#pragma unroll
            for (int j = 0; j < 8; ++j) afr.v[j] = (T)leaky((float)yfrag.v[j] * sc + sh, a.slope);
            const int tt = r < 9 ? r : 0;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int pix = wave * 64 + ks * 16 + 8 * h + j, ly = pix >> 5, lx = pix & 31;
                const float v = dl_s[(ly - tt / 3 + 2) * DW + (lx - tt % 3 + 2)];
                bfr.v[j] = (T)(r < 9 ? v : 0.f);
            }
            mma(accw, afr, bfr);
        }
        // ---- epilogue: dz = dA * leaky'(z) written in place over this wave's own y rows
#pragma unroll
        for (int mb = 0; mb < 2; ++mb) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int pix = (2 * wave + mb) * 32 + acc_row(i, lane);
                T* cell = reinterpret_cast<T*>(ytile + pix * PITCH) + r;
                // (explicit fused multiply-adds: convout_bwd_mfma_kernel and convout_step_mfma_kernel must round alike, and
                //  left to the compiler the contraction of these expressions came out differently in the two kernels)
                const float yv = (float)(*cell), z = __builtin_fmaf(yv, sc, sh);
                const float dzv = (float)(T)(z > 0.f ? acca[mb][i] : acca[mb][i] * a.slope);
                *cell = (T)dzv;
                s1 += dzv; s2 = __builtin_fmaf(dzv, __builtin_fmaf(yv, is, xm), s2);
            }
        }
        // the wave re-reads only its own 64 pixels (in-order LDS within a wave): 4 KiB = 4 chunks per lane
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (a.store_dz) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int id = lane + 64 * u, pix = wave * 64 + (id >> 2), qq = id & 3;
                const T8 v = *reinterpret_cast<const T8*>(ytile + pix * PITCH + qq * 16);
                const size_t g = (((size_t)b * a.H + y0 + (pix >> 5)) * a.W + x0 + (pix & 31)) * 32 + qq * 8;
                *reinterpret_cast<T8*>(a.dz + g) = v;
            }
        }
    }

    // ---- workgroup reductions: dW (rows = channel, lanes 0..8 = tap), statistics, sum of dlogit
    s1 += __shfl_xor(s1, 32, 64); s2 += __shfl_xor(s2, 32, 64);
    sdl = wave_sum(sdl);
    __syncthreads();
    if (r < 9) {
#pragma unroll
        for (int i = 0; i < 16; ++i) red[wave][r * 32 + acc_row(i, lane)] = accw[i];
    }
    if (h == 0) { red[wave][288 + r] = s1; red[wave][320 + r] = s2; }
    if (lane == 0) red[wave][352] = sdl;
    __syncthreads();
    for (int j = tid; j < 288; j += 256) a.slab[(size_t)blockIdx.x * 288 + j] = red[0][j] + red[1][j] + red[2][j] + red[3][j];
    if (tid < 64) unsafeAtomicAdd(&a.stat[stat_rep() * 64 + tid], (double)(red[0][288 + tid] + red[1][288 + tid] + red[2][288 + tid] + red[3][288 + tid]));
    if (tid == 64) unsafeAtomicAdd(a.dbias + stat_rep() * 8, (double)(red[0][352] + red[1][352] + red[2][352] + red[3][352]));
}

// ---------------------------------------------------------------------------
// Output conv forward AND backward in one pass over y7 (16-bit modes, the fused training step only: forward with
// train = 2 leaves the output conv to the backward).  Per 8x32 tile: the (8+4)x(32+4) patch of a = LeakyReLU(BN(y)) is
// staged once and feeds (1) the per-pixel tap products of the forward (logits on the tile + 1-pixel halo -> sigmoid,
// BCE, xhat, dlogit), (2) the input gradient dA = dl (*) w and (3) the weight gradient a^T dl - y7 is read once instead
// of twice (x1.69 halo instead of x1.33 + x1.0) and dlogit never goes to HBM.  Same arithmetic, element for element,
// as convout_fwd_mfma_kernel followed by convout_bwd_mfma_kernel (tests/test_parity_gpu.py checks bit-identity).
template <typename T> struct ConvOutStepArgs {
    const T* yf; const float* wt; const float* bias; const float* target;
    float* xhat; double* accum;                 // accum[rep*8 + 0] += BCE sum, [rep*8 + 2] += sum of dlogit (bias gradient)
    T* dz; float* slab; double* stat;           // stat: sum dz | sum dz*xhat7 of final_layer's BatchNorm (replicated)
    int B, H, W, n_tiles; float inv_n, slope, gmul;
    BnFuse fuse;                                // final_layer BatchNorm finalised in the prologue (forward mode)
    int rev;
    int ablate;                                 // timing diagnostics only (results wrong): 1 forward tap MFMAs, 2 logits, 4 gradient MFMAs, 8 epilogue, 16 dz store
};

static inline size_t convout_step_lds() { return 448 * 80 + 8 * 32 * 80 + 448 * 9 * 4 + (10 * 34 + 8) * 4 + 4 * (32 * 9 + 64 + 2) * 4 + 128 * 4; }

template <typename T>
__global__ __launch_bounds__(256, 2) void convout_step_mfma_kernel(ConvOutStepArgs<T> a) {
    typedef typename H16<T>::v8 T8;
    constexpr int TH = 8, TW = 32, PH = TH + 4, PW = TW + 4, NP = PH * PW, NPAD = 448, PITCH = 80, NCHK = NP * 4, MAXI = 7;
    constexpr int DH = TH + 2, DW = TW + 2, NDL = DH * DW;
    constexpr int RED = 32 * 9 + 64 + 2;
    extern __shared__ __attribute__((aligned(16))) char smem[];           // convout_step_lds() bytes (> the 64 KB static limit)
    char* atile = smem;                                                    // a = LeakyReLU(BN(y)) on the patch
    char* ytile = atile + NPAD * PITCH;                                    // raw y of the tile, dz in place
    float* part = reinterpret_cast<float*>(ytile + TH * TW * PITCH);       // [NPAD][9]
    float* dl_s = part + NPAD * 9;                                         // [NDL + 6]
    float (*red)[RED] = reinterpret_cast<float (*)[RED]>(dl_s + NDL + 8);  // [4][RED]
    float* cf = reinterpret_cast<float*>(red) + 4 * RED;                   // scale | shift | invstd | -mean*invstd
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, h = lane >> 5;
    const int g4 = lane >> 4, li = lane & 15, q = li >> 2, p = li & 3;
    const int tiles_x = a.W / TW, tiles_y = a.H / TH;
    if (tid < 32) { float k1; bn_fused_channel(a.fuse, tid, blockIdx.x == 0, cf[tid], k1, cf[32 + tid], &cf[64 + tid], &cf[96 + tid]); }
    for (int i = tid; i < (NPAD - NP) * PITCH / 16; i += 256) *reinterpret_cast<f32x4*>(atile + NP * PITCH + i * 16) = f32x4{0.f, 0.f, 0.f, 0.f};
    Frag<T> wf[2];       // forward: B[k = channel][col = tap r]
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int j = 0; j < 8; ++j) wf[ks].v[j] = (T)(r < 9 ? a.wt[r * 32 + ks * 16 + 8 * h + j] : 0.f);
    Frag<T> wfrag;       // input gradient: B[k = tap][col = channel r]
#pragma unroll
    for (int j = 0; j < 8; ++j) { const int t = 8 * h + j; wfrag.v[j] = (T)(t < 9 ? a.wt[t * 32 + r] : 0.f); }
    const float bo = a.bias[0], gs = a.gmul;
    float bsum = 0.f, s1 = 0.f, s2 = 0.f, sdl = 0.f;
    f32x16 accw;
#pragma unroll
    for (int i = 0; i < 16; ++i) accw[i] = 0.f;

    auto tile_origin = [&](int tile_, int& b, int& y0, int& x0) {
        const int tile = a.rev ? a.n_tiles - 1 - tile_ : tile_;
        const int tx = tile % tiles_x, ty = (tile / tiles_x) % tiles_y;
        b = tile / (tiles_x * tiles_y); y0 = ty * TH; x0 = tx * TW;
    };
    T8 pre[MAXI]; int ok[MAXI]; float pretg[2]; int tgok[2];
    auto prefetch = [&](int tile) {
        int b, y0, x0; tile_origin(tile, b, y0, x0);
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int i = tid + 256 * u, ry = i / DW, rx = i - ry * DW, gy = y0 - 1 + ry, gx = x0 - 1 + rx;
            tgok[u] = i < NDL && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
            pretg[u] = a.target[tgok[u] ? ((size_t)b * a.H + gy) * a.W + gx : 0];
        }
        const int base = ((b * a.H + y0 - 2) * a.W + x0 - 2) * 32 + (tid & 3) * 8;
#pragma unroll
        for (int u = 0; u < MAXI; ++u) {
            const int id = tid + 256 * u, pix = id >> 2;
            const int py = pix / PW, px = pix - py * PW, gy = y0 - 2 + py, gx = x0 - 2 + px;
            ok[u] = id < NCHK && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
            const uint32_t g = ok[u] ? (uint32_t)(base + (py * a.W + px) * 32) * 2u : 0u;   // byte offset (< 4 GiB: host check)
            pre[u] = *reinterpret_cast<const T8*>(reinterpret_cast<const char*>(a.yf) + g);
        }
    };
    __syncthreads();                     // cf published
    f32x2 kc[4], kh[4];                  // this thread's 8 staging channels: scale / shift pairs
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        kc[e] = f32x2{cf[(tid & 3) * 8 + 2 * e], cf[(tid & 3) * 8 + 2 * e + 1]};
        kh[e] = f32x2{cf[32 + (tid & 3) * 8 + 2 * e], cf[32 + (tid & 3) * 8 + 2 * e + 1]};
    }
    const float sc = cf[r], sh = cf[32 + r], is = cf[64 + r], xm = cf[96 + r];   // epilogue: channel r

    int tile = blockIdx.x;
    if (tile < a.n_tiles) prefetch(tile);
    for (; tile < a.n_tiles; tile += gridDim.x) {
        int b, y0, x0; tile_origin(tile, b, y0, x0);
        __syncthreads();   // previous tile fully consumed
#pragma unroll
        for (int u = 0; u < MAXI; ++u) {
            const int id = tid + 256 * u, pix = id >> 2;
            T8 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                f32x2 z = f32x2{(float)pre[u][2 * e], (float)pre[u][2 * e + 1]} * kc[e] + kh[e];
                const f32x2 zs = z * a.slope;
                z.x = fmaxf(z.x, zs.x); z.y = fmaxf(z.y, zs.y);
                o[2 * e] = (T)z.x; o[2 * e + 1] = (T)z.y;
            }
            if (!ok[u]) o = T8{0, 0, 0, 0, 0, 0, 0, 0};
            if (id < NCHK) {
                *reinterpret_cast<T8*>(atile + pix * PITCH + (id & 3) * 16) = o;
                const int py = pix / PW, px = pix - py * PW;
                if (py >= 2 && py < TH + 2 && px >= 2 && px < TW + 2)     // the tile itself: raw y for the epilogue
                    *reinterpret_cast<T8*>(ytile + ((py - 2) * TW + px - 2) * PITCH + (id & 3) * 16) = pre[u];
            }
        }
        const float tg0 = pretg[0], tg1 = pretg[1]; const int tk0 = tgok[0], tk1 = tgok[1];
        __syncthreads();
        if (tile + (int)gridDim.x < a.n_tiles) prefetch(tile + gridDim.x);   // in flight during everything below
        // ---- forward tap products: 14 row blocks of 32 patch pixels
        if (!VAE_ABLATE(a.ablate, 1))
        for (int mb = wave; mb < NPAD / 32; mb += 4) {
            const int row0 = mb * 32;
            f32x16 acc;
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                Frag<T> af = load_frag(reinterpret_cast<const T*>(atile + (row0 + r) * PITCH + ks * 32) + h * 8);
                mma(acc, af, wf[ks]);
            }
            if (r < 9) {
#pragma unroll
                for (int i = 0; i < 16; ++i) part[(row0 + acc_row(i, lane)) * 9 + r] = acc[i];
            }
        }
        __syncthreads();
        // ---- logits, sigmoid, BCE and dlogit on the tile + 1-pixel halo (340 pixels)
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int i = tid + 256 * u;
            if (i < NDL && !VAE_ABLATE(a.ablate, 2)) {
                const int ry = i / DW, rx = i - ry * DW;
                float logit = bo;
#pragma unroll
                for (int t = 0; t < 9; ++t) logit += part[((ry + t / 3) * PW + rx + t % 3) * 9 + t];
                const float tg = u ? tg1 : tg0;
                const float xh = 1.f / (1.f + expf(-logit));
                const float om = xh * (1.f - xh);
                const float dlv = (xh - tg) / fmaxf(om, 1e-12f) * om * a.inv_n;
                const float dl = (u ? tk1 : tk0) ? dlv * gs : 0.f;
                dl_s[i] = dl;
                if (ry >= 1 && ry <= TH && rx >= 1 && rx <= TW) {   // the tile itself
                    const float l1 = fmaxf(logf(xh), -100.f), l0 = fmaxf(logf(1.f - xh), -100.f);
                    bsum += -(tg * l1 + (1.f - tg) * l0);
                    a.xhat[((size_t)b * a.H + y0 + ry - 1) * a.W + x0 + rx - 1] = xh;
                    sdl += dl;
                }
            }
        }
        __syncthreads();
        // ---- dA: 2 blocks of 32 pixels per wave (tile rows 2*wave, 2*wave+1)
        f32x16 acca[2];
#pragma unroll
        for (int mb = 0; mb < 2; ++mb) {
#pragma unroll
            for (int i = 0; i < 16; ++i) acca[mb][i] = 0.f;
            if (VAE_ABLATE(a.ablate, 4)) continue;
            const int ly = 2 * wave + mb, lx = r;
            Frag<T> af;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int t = 8 * h + j, tt = t < 9 ? t : 0;
                const float v = dl_s[(ly - tt / 3 + 2) * DW + (lx - tt % 3 + 2)];
                af.v[j] = (T)(t < 9 ? v : 0.f);
            }
            mma(acca[mb], af, wfrag);
        }
        // ---- dW: K = the wave's 64 pixels, A = a^T (k-major via tr16 from the staged patch), B = dl taps
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            if (VAE_ABLATE(a.ablate, 4)) continue;
            const int k0 = wave * 64 + ks * 16 + 8 * (g4 >> 1) + q;            // tile pixel; k0 + 4 is in the same tile row
            const int row = ((k0 >> 5) + 2) * PW + (k0 & 31) + 2;
            const int col = (16 * (g4 & 1) + 4 * p) * 2;
            Frag<T> afr = frag_tr16<T>(atile + row * PITCH + col, atile + (row + 4) * PITCH + col);
            Frag<T> bfr;
            const int tt = r < 9 ? r : 0;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int pix = wave * 64 + ks * 16 + 8 * h + j, ly = pix >> 5, lx = pix & 31;
                const float v = dl_s[(ly - tt / 3 + 2) * DW + (lx - tt % 3 + 2)];
                bfr.v[j] = (T)(r < 9 ? v : 0.f);
            }
            mma(accw, afr, bfr);
        }
        // ---- epilogue: dz = dA * leaky'(z) written in place over this wave's own y rows
        if (!VAE_ABLATE(a.ablate, 8))
#pragma unroll
        for (int mb = 0; mb < 2; ++mb) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int pix = (2 * wave + mb) * 32 + acc_row(i, lane);
                T* cell = reinterpret_cast<T*>(ytile + pix * PITCH) + r;
                // (explicit fused multiply-adds: convout_bwd_mfma_kernel and convout_step_mfma_kernel must round alike, and
                //  left to the compiler the contraction of these expressions came out differently in the two kernels)
                const float yv = (float)(*cell), z = __builtin_fmaf(yv, sc, sh);
                const float dzv = (float)(T)(z > 0.f ? acca[mb][i] : acca[mb][i] * a.slope);
                *cell = (T)dzv;
                s1 += dzv; s2 = __builtin_fmaf(dzv, __builtin_fmaf(yv, is, xm), s2);
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the wave re-reads only its own 64 pixels
        if (!VAE_ABLATE(a.ablate, 16))
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int id = lane + 64 * u, pix = wave * 64 + (id >> 2), qq = id & 3;
            const T8 v = *reinterpret_cast<const T8*>(ytile + pix * PITCH + qq * 16);
            const size_t g = (((size_t)b * a.H + y0 + (pix >> 5)) * a.W + x0 + (pix & 31)) * 32 + qq * 8;
            *reinterpret_cast<T8*>(a.dz + g) = v;
        }
    }

    // ---- workgroup reductions: dW (rows = channel, lanes 0..8 = tap), statistics, sum of dlogit, BCE sum
    s1 += __shfl_xor(s1, 32, 64); s2 += __shfl_xor(s2, 32, 64);
    sdl = wave_sum(sdl); bsum = wave_sum(bsum);
    __syncthreads();
    if (r < 9) {
#pragma unroll
        for (int i = 0; i < 16; ++i) red[wave][r * 32 + acc_row(i, lane)] = accw[i];
    }
    if (h == 0) { red[wave][288 + r] = s1; red[wave][320 + r] = s2; }
    if (lane == 0) { red[wave][352] = sdl; red[wave][353] = bsum; }
    __syncthreads();
    for (int j = tid; j < 288; j += 256) a.slab[(size_t)blockIdx.x * 288 + j] = red[0][j] + red[1][j] + red[2][j] + red[3][j];
    if (tid < 64) unsafeAtomicAdd(&a.stat[stat_rep() * 64 + tid], (double)(red[0][288 + tid] + red[1][288 + tid] + red[2][288 + tid] + red[3][288 + tid]));
    if (tid == 64) unsafeAtomicAdd(&a.accum[stat_rep() * 8 + 2], (double)(red[0][352] + red[1][352] + red[2][352] + red[3][352]));
    if (tid == 65) unsafeAtomicAdd(&a.accum[stat_rep() * 8 + 0], (double)(red[0][353] + red[1][353] + red[2][353] + red[3][353]));
}

// ---------------------------------------------------------------------------
template <typename T> struct DenseArgs {
    const T* A; const float* coef; float slope; int C;  // transform channel = k & (C-1); coef==nullptr -> identity
    const T* Bp;                                        // packed [K/8][Npad][8]
    float* slab;                                        // [nsplit][M][Npad]
    int M, K, Npad, ksteps_per_split;
    BnFuse fuse;                                        // mode == BNF_FWD: coefficients derived from batch statistics (C <= 256)
};

template <typename T, int NT>
__global__ __launch_bounds__(256) void dense_kernel(DenseArgs<T> a) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, h = lane >> 5;
    const int m = blockIdx.x * 128 + wave * 32 + r, n0 = blockIdx.z * 32 * NT;
    const int ks0 = blockIdx.y * a.ksteps_per_split, ks1 = min(a.K / 16, ks0 + a.ksteps_per_split);
    __shared__ float dcf[2 * 256];   // scale | shift per channel
    if (a.coef) {
        const bool writer = blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0;
        for (int c = tid; c < a.C; c += 256) {
            if (a.fuse.mode == BNF_FWD) { float k1; bn_fused_channel(a.fuse, c, writer, dcf[c], k1, dcf[256 + c]); }
            else { dcf[c] = a.coef[c]; dcf[256 + c] = a.coef[2 * a.C + c]; }
        }
        __syncthreads();
    }
    f32x16 acc[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[nt][i] = 0.f;
    for (int ks = ks0; ks < ks1; ++ks) {
        const int k = ks * 16 + 8 * h;
        Frag<T> af;
        if (m < a.M) {
            Frag<T> raw = load_frag(a.A + (size_t)m * a.K + k);
            if (a.coef) {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int c = (k + j) & (a.C - 1);
                    float v;
                    if constexpr (sizeof(T) == 4) v = raw.v[j]; else v = (float)raw.v[j];
                    af.set(j, leaky(v * dcf[c] + dcf[256 + c], a.slope));
                }
            } else {
                af = raw;
            }
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) af.set(j, 0.f);
        }
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            // columns beyond Npad (only if a launcher ever picks an NT that does not divide Npad/32) read column 0 and are dropped below
            const int n = n0 + nt * 32 + r;
            Frag<T> bf = load_frag(a.Bp + ((size_t)(k >> 3) * a.Npad + (n < a.Npad ? n : 0)) * 8);
            mma(acc[nt], af, bf);
        }
    }
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int mm = blockIdx.x * 128 + wave * 32 + acc_row(i, lane), n = n0 + nt * 32 + r;
            if (mm < a.M && n < a.Npad) a.slab[((size_t)blockIdx.y * a.M + mm) * a.Npad + n] = acc[nt][i];
        }
}
