// The skinny linears around the latent (models.py:137-141, 162 of the reference and their gradients) on the exact-f32 MFMA
// (v_mfma_f32_32x32x2_f32: a k-ordered fmaf chain, so every storage type keeps f32 arithmetic) for gfx950.
//
// Every one of these products has ONE huge dimension (F = 256*(H/16)^2 features) and two small ones (batch B, latent L).  The
// round-1/2 kernels gave a thread one feature column and streamed the bottleneck tensor with 2-byte loads (0.3-0.8 TB/s on an
// 8 MB tensor, split over the batch into slabs that four extra launches summed).  Here a workgroup owns 64 feature columns for
// the WHOLE batch: the tensor tile comes in with 16-byte loads, goes through LDS once, and the contraction over the batch
// (weight gradients) or over the latent (input gradient / decoder_input forward) runs on the matrix pipe - no split over the
// batch, no slabs, no reduction launches:
//   batch_gemm_kernel   OUT[f][n] = sum_b X[b][f] * Y[b][n]     fc_mu|fc_var weight gradient (X = LeakyReLU(BN(y3)), Y = dlat)
//                                                                decoder_input weight + bias gradient (X = dd0, Y = z | 1)
//   row_gemm_kernel     OUT[b][f] = sum_k Y[b][k] * W[k][f]     decoder_input forward (Y = z, + bias) and the fc input gradient
//                                                                (Y = dlat, LeakyReLU'/BatchNorm-statistics epilogue of encoder.3)
// A tile's 64 feature columns are P pixels x 64/P channels of the NHWC bottleneck (P = min(s2, 8)), enumerated pixel-fastest, so
// that consecutive columns are consecutive in the reference's NCHW flatten order (models.py:133: f = c*s2 + pix) - the order of
// the weight matrices' F axis - while every global access of the tensor itself is a 16-byte run of 8 channels.
#pragma once
#include "common.cuh"

struct FeatTile {   // geometry of the 64-column feature tile of workgroup `blk` (every quantity a power of two: shifts, no integer division)
    int P, lP, CW, pix0, c0, s2;
    __device__ __forceinline__ FeatTile(int blk, int s2_) : s2(s2_) {
        P = s2_ < 8 ? s2_ : 8; lP = 31 - __clz(P); CW = 64 >> lP;
        const int lpb = (31 - __clz(s2_)) - lP;                   // log2 of the pixel blocks per channel group
        const int cblk = blk >> lpb; pix0 = (blk - (cblk << lpb)) << lP; c0 = cblk * CW;
    }
    // column = cl * P + p  (cl: channel within the tile, p: pixel within the tile)
    __device__ __forceinline__ int fp(int col) const { return (pix0 + (col & (P - 1))) * 256 + c0 + (col >> lP); }     // NHWC flatten index
    __device__ __forceinline__ int fref(int col) const { return (c0 + (col >> lP)) * s2 + pix0 + (col & (P - 1)); }    // reference flatten index
    __device__ __forceinline__ int chan(int col) const { return c0 + (col >> lP); }
};

// Stage rows [b0, b0+256) x the tile's 64 columns of X (NHWC [B][F] of T) into LDS as f32 [256][64] (column order of FeatTile),
// applying v -> leaky(v*sc[c] + sh[c]) when coef != nullptr (coef rows: LC_SC at 0, LC_SH at 2*256).  Rows beyond B are zero.
template <typename T>
__device__ __forceinline__ void stage_feat_tile_f32(const T* __restrict__ X, const FeatTile& t, int b0, int B, int F, const float* __restrict__ coef,
                                                    float slope, float* xs, int tid) {
    constexpr int E = 16 / sizeof(T);                 // elements per 16-byte vector
    constexpr int vpr = 64 / E;                       // vectors per row
    const int lgpp = (6 - t.lP) - (E == 8 ? 3 : 2);   // log2 of the vectors per pixel (CW / E)
    for (int v = tid; v < 256 * vpr; v += 256) {
        const int row = v / vpr, j = v - row * vpr, p = j >> lgpp, g = j - (p << lgpp);
        const int b = b0 + row;
        float o[E];
        if (b < B) {
            const Vec16<T> raw = *reinterpret_cast<const Vec16<T>*>(X + (size_t)b * F + (t.pix0 + p) * 256 + t.c0 + g * E);
#pragma unroll
            for (int e = 0; e < E; ++e) {
                float x = raw.get(e);
                if (coef) { const int c = t.c0 + g * E + e; x = leaky(x * coef[c] + coef[2 * 256 + c], slope); }
                o[e] = x;
            }
        } else {
#pragma unroll
            for (int e = 0; e < E; ++e) o[e] = 0.f;
        }
#pragma unroll
        for (int e = 0; e < E; ++e) xs[row * 64 + (g * E + e) * t.P + p] = o[e];
    }
}

struct BatchGemmArgs {
    const void* X; const float* coef; float slope;   // X [B][F] of T; coef: LeakyReLU(BN) staging coefficients of encoder.3, or nullptr (identity)
    const float* Y; int ldy, ncols;                  // Y [B][ldy] f32, columns 0..ncols-1
    int ones_col;                                    // 1: an extra column ncols of ones (bias gradient)
    float* out0; float* out1; float* outb;           // TRANSPOSED (fc): out0 = dW_mu [L][F], out1 = dW_var [L][F] (rows n < L / >= L)
                                                     // else (decoder_input): out0 = dWd [F][L], outb = dbd [F] (the ones column)
    float* colsum0; float* colsum1;                  // fc: column sums of Y -> bias gradients (workgroup 0)
    int B, F, L, s2; float scale;                    // scale: every output is written times it (f16 gradient scaling)
};

// OUT[f][n] = sum_b X[b][f] Y[b][n].  256 threads; waves = (feature half w & 1) x (batch half w >> 1 of each 256-row chunk); the two
// batch halves are added through LDS at the end.  TRANSPOSED: the product is formed as OUT^T (rows = n, lanes = f) so that the
// stores of [n][F]-major outputs run along consecutive reference feature indices.
template <typename T, bool TRANSPOSED>
__global__ __launch_bounds__(256) void batch_gemm_kernel(BatchGemmArgs a) {
    constexpr int NBMAX = 9;                          // up to 9 x 32 columns of Y (L = 128: 256 + none / 128 + ones)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* xs = reinterpret_cast<float*>(smem);       // [256][64]
    float* ys = xs + 256 * 64;                        // [256][32]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, h = lane >> 5;
    const int fblk = wave & 1, khalf = wave >> 1;
    const FeatTile t(blockIdx.x, a.s2);
    const int ntot = a.ncols + (a.ones_col ? 1 : 0), nblk = (ntot + 31) / 32;
    f32x16 acc[NBMAX];
#pragma unroll
    for (int nb = 0; nb < NBMAX; ++nb)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[nb][i] = 0.f;
    for (int b0 = 0; b0 < a.B; b0 += 256) {
        __syncthreads();                              // previous chunk consumed
        stage_feat_tile_f32<T>(reinterpret_cast<const T*>(a.X), t, b0, a.B, a.F, a.coef, a.slope, xs, tid);
#pragma unroll
        for (int nb = 0; nb < NBMAX; ++nb) {
            if (nb < nblk) {
                __syncthreads();                      // ys free (and, first block: xs published below)
                if ((a.ldy & 3) == 0 && (a.ncols & 3) == 0) {   // rows of Y 16-byte aligned: float4 loads
                    for (int i = tid; i < 256 * 8; i += 256) {
                        const int row = i >> 3, nq = (i & 7) * 4, n = nb * 32 + nq, b = b0 + row;
                        f32x4 v = {0.f, 0.f, 0.f, 0.f};
                        if (b < a.B) {
                            if (n < a.ncols) v = *reinterpret_cast<const f32x4*>(a.Y + (size_t)b * a.ldy + n);
                            else if (a.ones_col && n == a.ncols) v[0] = 1.f;
                        }
                        *reinterpret_cast<f32x4*>(ys + row * 32 + nq) = v;
                    }
                } else {
                    for (int i = tid; i < 256 * 32; i += 256) {
                        const int row = i >> 5, n = nb * 32 + (i & 31), b = b0 + row;
                        float v = 0.f;
                        if (b < a.B) v = n < a.ncols ? a.Y[(size_t)b * a.ldy + n] : ((a.ones_col && n == a.ncols) ? 1.f : 0.f);
                        ys[i] = v;
                    }
                }
                __syncthreads();
                if (blockIdx.x == 0 && a.colsum0) {   // bias gradients of fc_mu / fc_var: column sums of dlat (workgroup 0 only: uniform branch)
                    float s = 0.f;
                    for (int row = tid >> 5; row < 256; row += 8) s += ys[row * 32 + (tid & 31)];
                    float* cs = ys + 256 * 32;         // [8][32] scratch behind the tile
                    cs[tid] = s;
                    __syncthreads();
                    const int n = nb * 32 + tid;
                    if (tid < 32 && n < a.ncols) {
                        float v = 0.f;
#pragma unroll
                        for (int g = 0; g < 8; ++g) v += cs[g * 32 + tid];
                        float* dst = n < a.L ? a.colsum0 + n : a.colsum1 + (n - a.L);
                        *dst = (b0 == 0 ? 0.f : *dst) + v * a.scale;
                    }
                }
                const float* xp = xs + (khalf * 128 + h) * 64 + fblk * 32 + r;
                const float* yp = ys + (khalf * 128 + h) * 32 + r;
#pragma unroll 8
                for (int s = 0; s < 64; ++s) {        // k = batch rows 2s + h of this wave's half
                    const float xv = xp[s * 2 * 64], yv = yp[s * 2 * 32];
                    if constexpr (TRANSPOSED) acc[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(yv, xv, acc[nb], 0, 0, 0);
                    else acc[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(xv, yv, acc[nb], 0, 0, 0);
                }
            }
        }
    }
    // add the two batch halves (waves 2, 3 -> LDS -> waves 0, 1) and store
    __syncthreads();
    float* red = xs;                                   // [2 feature halves][nblk][16][64 lanes]
#pragma unroll
    for (int nb = 0; nb < NBMAX; ++nb)
        if (nb < nblk && khalf == 1)
#pragma unroll
            for (int i = 0; i < 16; ++i) red[((fblk * NBMAX + nb) * 16 + i) * 64 + lane] = acc[nb][i];
    __syncthreads();
    if (khalf == 0) {
#pragma unroll
        for (int nb = 0; nb < NBMAX; ++nb) {
            if (nb < nblk) {
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const float v = (acc[nb][i] + red[((fblk * NBMAX + nb) * 16 + i) * 64 + lane]) * a.scale;
                    if constexpr (TRANSPOSED) {        // rows = n, lanes = feature column
                        const int n = nb * 32 + acc_row(i, lane), col = fblk * 32 + r;
                        if (n < a.ncols) {
                            float* dst = n < a.L ? a.out0 + (size_t)n * a.F : a.out1 + (size_t)(n - a.L) * a.F;
                            dst[t.fref(col)] = v;
                        }
                    } else {                           // rows = feature column, lanes = n
                        const int col = fblk * 32 + acc_row(i, lane), n = nb * 32 + r;
                        if (n < a.ncols) a.out0[(size_t)t.fref(col) * a.L + n] = v;
                        else if (a.ones_col && n == a.ncols) a.outb[t.fref(col)] = v;
                    }
                }
            }
        }
    }
}
static inline size_t batch_gemm_lds() { return (size_t)(256 * 64 + 256 * 32 + 8 * 32) * 4; }

// ---------------------------------------------------------------------------------------------------------------
struct RowGemmArgs {
    const float* Y; int ldy, K;                       // Y [B][ldy] f32, K columns contracted
    const float* Wf; const void* Wp; int npad;        // weights: Wf f32 [F_ref][K] (decoder_input.weight), or Wp: packed T image [F/8][npad][8] indexed by the NHWC feature
    const float* bias;                                // EPI 0: [F_ref]
    void* out;                                        // EPI 0: d0 [B][F] of T; EPI 1: dz [B][F] of T
    const void* y; const float* ocoef; float slope;   // EPI 1: raw encoder.3 output and its coefficient block (LC_* rows, 256 channels)
    const float* gpre; float gmul;                    // EPI 1: optional gradient on pre_latents [B][F_ref]
    double* stat;                                     // EPI 1: [STAT_R][2][256]
    int B, F, s2;
};

// OUT[b][f] = sum_k Y[b][k] W[k][f] on a 64-column feature tile, all batch rows; K walked in chunks of 32.  EPI 0: + bias, stored as T.
// EPI 1 (fc input gradient): dz = LeakyReLU'(BN(y)) * OUT, BatchNorm-backward statistics sum dz, sum dz*xhat (models.py:137-141 backward).
template <typename T, int EPI>
__global__ __launch_bounds__(256) void row_gemm_kernel(RowGemmArgs a) {
    constexpr int KC = 32, YP = KC + 1;
    constexpr int E = 16 / sizeof(T);
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* ws = reinterpret_cast<float*>(smem);        // [KC][64]
    float* ys = ws + KC * 64;                          // [256][YP]
    T* ot = reinterpret_cast<T*>(ys + 256 * YP);       // [256][64] out tile (EPI 1: preloaded with y)
    float* sred = reinterpret_cast<float*>(ot + 256 * 64);   // [4 waves][2][32]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, h = lane >> 5;
    const int fblk = wave & 1, bq = wave >> 1;         // wave: feature half, and batch blocks bq*4 .. bq*4+3 of each 256-row chunk
    const FeatTile t(blockIdx.x, a.s2);
    const int col = fblk * 32 + r;
    float bv = 0.f, sc = 0.f, sh = 0.f, is = 0.f, xm = 0.f, s1 = 0.f, s2 = 0.f;
    if constexpr (EPI == 0) bv = a.bias ? a.bias[t.fref(col)] : 0.f;
    else {
        const int c = t.chan(col);
        sc = a.ocoef[LC_SC * 256 + c]; sh = a.ocoef[LC_SH * 256 + c]; is = a.ocoef[LC_INVSTD * 256 + c]; xm = a.ocoef[LC_XM * 256 + c];
    }
    constexpr int vpr = 64 / E;
    const int lgpp = (6 - t.lP) - (E == 8 ? 3 : 2);
    for (int b0 = 0; b0 < a.B; b0 += 256) {
        f32x16 acc[4];
#pragma unroll
        for (int bb = 0; bb < 4; ++bb)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[bb][i] = bv;
        for (int k0 = 0; k0 < a.K; k0 += KC) {
            __syncthreads();                           // previous chunk / previous batch chunk's tile consumed
            if (a.Wf && (a.K & 3) == 0) {                 // f32 weights [F_ref][K], rows 16-byte aligned: one float4 per (column, 4 k)
                for (int i = tid; i < 64 * (KC / 4); i += 256) {
                    const int cc = i >> 3, kq = (i & 7) * 4, k = k0 + kq;
                    f32x4 v = {0.f, 0.f, 0.f, 0.f};
                    if (k < a.K) v = *reinterpret_cast<const f32x4*>(a.Wf + (size_t)t.fref(cc) * a.K + k);
#pragma unroll
                    for (int e = 0; e < 4; ++e) ws[(kq + e) * 64 + cc] = v[e];
                }
            } else {
                for (int i = tid; i < KC * 64; i += 256) {   // weights of the tile for this K chunk
                    const int k = k0 + (i >> 6), cc = i & 63;
                    float w = 0.f;
                    if (k < a.K) {
                        if (a.Wf) w = a.Wf[(size_t)t.fref(cc) * a.K + k];
                        else { const int f = t.fp(cc); w = tofloat(reinterpret_cast<const T*>(a.Wp)[((size_t)(f >> 3) * a.npad + k) * 8 + (f & 7)]); }
                    }
                    ws[i] = w;
                }
            }
            if ((a.ldy & 3) == 0 && (a.K & 3) == 0) {      // rows of Y are 16-byte aligned: float4 loads
                for (int i = tid; i < 256 * (KC / 4); i += 256) {
                    const int row = i >> 3, kq = (i & 7) * 4, k = k0 + kq, b = b0 + row;
                    f32x4 v = {0.f, 0.f, 0.f, 0.f};
                    if (b < a.B && k < a.K) v = *reinterpret_cast<const f32x4*>(a.Y + (size_t)b * a.ldy + k);
#pragma unroll
                    for (int e = 0; e < 4; ++e) ys[row * YP + kq + e] = v[e];
                }
            } else {
                for (int i = tid; i < 256 * KC; i += 256) {
                    const int row = i >> 5, k = k0 + (i & 31), b = b0 + row;
                    ys[row * YP + (i & 31)] = (b < a.B && k < a.K) ? a.Y[(size_t)b * a.ldy + k] : 0.f;
                }
            }
            if (EPI == 1 && k0 == 0) {                 // y rows of the tile, 16-byte loads
                for (int v = tid; v < 256 * vpr; v += 256) {
                    const int row = v / vpr, j = v - row * vpr, p = j >> lgpp, g = j - (p << lgpp), b = b0 + row;
                    Vec16<T> raw = zero_vec16<T>();
                    if (b < a.B) raw = *reinterpret_cast<const Vec16<T>*>(reinterpret_cast<const T*>(a.y) + (size_t)b * a.F + (t.pix0 + p) * 256 + t.c0 + g * E);
#pragma unroll
                    for (int e = 0; e < E; ++e) ot[row * 64 + (g * E + e) * t.P + p] = fromfloat<T>(raw.get(e));
                }
            }
            __syncthreads();
            const int ksteps = (min(KC, a.K - k0) + 1) >> 1;
#pragma unroll
            for (int bb = 0; bb < 4; ++bb) {
                const float* yp = ys + ((bq * 4 + bb) * 32 + r) * YP + h;
                const float* wp = ws + h * 64 + col;
                for (int s = 0; s < ksteps; ++s) acc[bb] = __builtin_amdgcn_mfma_f32_32x32x2f32(yp[2 * s], wp[2 * s * 64], acc[bb], 0, 0, 0);
            }
        }
#pragma unroll
        for (int bb = 0; bb < 4; ++bb) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int row = (bq * 4 + bb) * 32 + acc_row(i, lane), b = b0 + row;
                if constexpr (EPI == 0) ot[row * 64 + col] = fromfloat<T>(acc[bb][i]);
                else if (b < a.B) {
                    float da = acc[bb][i];
                    if (a.gpre) da += a.gpre[(size_t)b * a.F + t.fref(col)] * a.gmul;
                    const float yv = tofloat(ot[row * 64 + col]);
                    const float z = yv * sc + sh;
                    const float dzv = round_as<T>(z > 0.f ? da : da * a.slope);
                    ot[row * 64 + col] = fromfloat<T>(dzv);
                    s1 += dzv; s2 += dzv * (yv * is + xm);
                }
            }
        }
        __syncthreads();
        // the tile goes out as 16-byte runs of 8 channels
        for (int v = tid; v < 256 * vpr; v += 256) {
            const int row = v / vpr, j = v - row * vpr, p = j >> lgpp, g = j - (p << lgpp), b = b0 + row;
            if (b < a.B) {
                Vec16<T> o;
#pragma unroll
                for (int e = 0; e < E; ++e) o.set(e, tofloat(ot[row * 64 + (g * E + e) * t.P + p]));
                *reinterpret_cast<Vec16<T>*>(reinterpret_cast<T*>(a.out) + (size_t)b * a.F + (t.pix0 + p) * 256 + t.c0 + g * E) = o;
            }
        }
    }
    if constexpr (EPI == 1) {
        // per-channel statistics: a channel owns P adjacent columns; lanes r and r + 32 hold the same column
        s1 += __shfl_xor(s1, 32, 64); s2 += __shfl_xor(s2, 32, 64);
        for (int o = 1; o < t.P; o <<= 1) { s1 += __shfl_xor(s1, o, 64); s2 += __shfl_xor(s2, o, 64); }
        __syncthreads();
        if (h == 0) { sred[(wave * 2 + 0) * 32 + r] = s1; sred[(wave * 2 + 1) * 32 + r] = s2; }
        __syncthreads();
        if (tid < 64 && (tid % t.P) == 0) {            // column tid: first pixel of its channel; waves fblk and fblk + 2 share the column half
            const int fb = tid >> 5, rr = tid & 31, c = t.chan(tid);
            const float v1 = sred[(fb * 2 + 0) * 32 + rr] + sred[((fb + 2) * 2 + 0) * 32 + rr];
            const float v2 = sred[(fb * 2 + 1) * 32 + rr] + sred[((fb + 2) * 2 + 1) * 32 + rr];
            double* st_ = a.stat + stat_rep() * 512;
            unsafeAtomicAdd(&st_[c], (double)v1);
            unsafeAtomicAdd(&st_[256 + c], (double)v2);
        }
    }
}
template <typename T> static inline size_t row_gemm_lds() { return (size_t)(32 * 64 + 256 * 33) * 4 + (size_t)256 * 64 * sizeof(T) + 4 * 2 * 32 * 4; }
