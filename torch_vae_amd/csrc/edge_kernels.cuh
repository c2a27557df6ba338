// Non-GEMM-shaped kernels of the VAE step for gfx950: the 1-channel ends of the
// network (HBM streams), the latent block, BatchNorm finalisation, weight packing
// and the fused AdamW update.  All reductions: registers -> wave shuffles -> LDS
// -> one double-precision atomic per channel per workgroup.
#pragma once
#include "common.cuh"

// reduce v across the lanes that share (lane & 3): offsets 4..32
__device__ __forceinline__ float cg_sum(float v) {
#pragma unroll
    for (int o = 4; o < 64; o <<= 1) v += __shfl_xor(v, o, 64);
    return v;
}

template <typename T> __device__ __forceinline__ void load8(const T* p, float* o);
template <> __device__ __forceinline__ void load8<float>(const float* p, float* o) {
    f32x4 a = *reinterpret_cast<const f32x4*>(p), b = *reinterpret_cast<const f32x4*>(p + 4);
#pragma unroll
    for (int i = 0; i < 4; ++i) { o[i] = a[i]; o[4 + i] = b[i]; }
}
template <> __device__ __forceinline__ void load8<bf16>(const bf16* p, float* o) {
    bf16x8 a = *reinterpret_cast<const bf16x8*>(p);
#pragma unroll
    for (int i = 0; i < 8; ++i) o[i] = (float)a[i];
}
template <> __device__ __forceinline__ void load8<f16>(const f16* p, float* o) {
    f16x8 a = *reinterpret_cast<const f16x8*>(p);
#pragma unroll
    for (int i = 0; i < 8; ++i) o[i] = (float)a[i];
}
template <typename T> __device__ __forceinline__ void store8(T* p, const float* o);
template <> __device__ __forceinline__ void store8<float>(float* p, const float* o) {
    f32x4 a, b;
#pragma unroll
    for (int i = 0; i < 4; ++i) { a[i] = o[i]; b[i] = o[4 + i]; }
    *reinterpret_cast<f32x4*>(p) = a; *reinterpret_cast<f32x4*>(p + 4) = b;
}
template <> __device__ __forceinline__ void store8<bf16>(bf16* p, const float* o) {
    bf16x8 a;
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = (bf16)o[i];
    *reinterpret_cast<bf16x8*>(p) = a;
}

template <> __device__ __forceinline__ void store8<f16>(f16* p, const float* o) {
    f16x8 a;
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = (f16)o[i];
    *reinterpret_cast<f16x8*>(p) = a;
}

// ---------------------------------------------------------------------------
// encoder block 0: Conv2d(1->32,k3,s2,p1) forward (models.py:45 with in_channels=1).
// x [B,H,W] f32 -> y [B,H/2,W/2,32] T, plus sum / sum-of-squares per channel.
// thread = (pixel slot, 8-channel group); a pixel's 32 channels are 4 adjacent lanes.
template <typename T>
__global__ __launch_bounds__(256) void conv1_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                        const float* __restrict__ bias, T* __restrict__ y,
                                                        double* __restrict__ stat, int B, int H, int W) {
    __shared__ float red[4][4][16];
    const int tid = threadIdx.x, cg = tid & 3, slot = tid >> 2, lane = tid & 63, wave = tid >> 6;
    const int Ho = H >> 1, Wo = W >> 1, lw = 31 - __builtin_clz(Wo), lh = 31 - __builtin_clz(Ho);
    const int P = B * Ho * Wo;
    float wr[8][9], br[8], s1[8], s2[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        br[c] = bias[cg * 8 + c]; s1[c] = 0.f; s2[c] = 0.f;
#pragma unroll
        for (int t = 0; t < 9; ++t) wr[c][t] = w[(cg * 8 + c) * 9 + t];
    }
    // A thread computes 4 consecutive output pixels of one row (8 channels each): their 3x9 input window is two aligned
    // float4 loads + one scalar per row instead of 36 scalar loads with bounds checks (W is a multiple of 8, so only
    // the row above the image and the column left of it can fall outside).
    const int Wq = Wo >> 2, lq = lw - 2, Q = B * Ho * Wq;
    for (int q = blockIdx.x * 64 + slot; q < Q; q += gridDim.x * 64) {
        const int oxq = q & (Wq - 1), oy = (q >> lq) & (Ho - 1), b = q >> (lq + lh);
        const int ix0 = 8 * oxq;
        float win[3][9];
#pragma unroll
        for (int rr = 0; rr < 3; ++rr) {
            const int iy = 2 * oy + rr - 1;
            const bool rin = iy >= 0;
            const float* row = x + ((size_t)b * H + (rin ? iy : 0)) * W + ix0;
            const f32x4 v0 = *reinterpret_cast<const f32x4*>(row), v1 = *reinterpret_cast<const f32x4*>(row + 4);
            const float left = (ix0 > 0) ? row[-1] : 0.f;
            win[rr][0] = rin ? left : 0.f;
#pragma unroll
            for (int e = 0; e < 4; ++e) { win[rr][1 + e] = rin ? v0[e] : 0.f; win[rr][5 + e] = rin ? v1[e] : 0.f; }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            float o[8];
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                float acc = br[c];
#pragma unroll
                for (int t = 0; t < 9; ++t) acc += wr[c][t] * win[t / 3][2 * u + t % 3];
                o[c] = round_as<T>(acc); s1[c] += o[c]; s2[c] += o[c] * o[c];
            }
            const size_t p = ((size_t)b * Ho + oy) * Wo + 4 * oxq + u;
            store8<T>(y + p * 32 + cg * 8, o);
        }
    }
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        s1[c] = cg_sum(s1[c]); s2[c] = cg_sum(s2[c]);
        if (lane < 4) { red[wave][lane][c] = s1[c]; red[wave][lane][8 + c] = s2[c]; }
    }
    __syncthreads();
    if (tid < 64) {
        const int ch = tid & 31, k = tid >> 5;  // k=0: sum, k=1: sumsq
        float s = 0.f;
#pragma unroll
        for (int wv = 0; wv < 4; ++wv) s += red[wv][ch >> 3][k * 8 + (ch & 7)];
        unsafeAtomicAdd(&stat[stat_rep() * 64 + k * 32 + ch], (double)s);
    }
}

// encoder block 0 weight gradient: dW[co][t] = sum_p G[p][co] * x[p (+) t], G = dz*p0 + y*p1 + p2
template <typename T>
__global__ __launch_bounds__(256) void conv1_wgrad_kernel(const float* __restrict__ x, const T* __restrict__ dz,
                                                          const T* __restrict__ y, const float* __restrict__ gcoef,
                                                          float* __restrict__ slab, int B, int H, int W, BnFuse fuse) {
    __shared__ float red[4][4][72];
    const int tid = threadIdx.x, cg = tid & 3, slot = tid >> 2, lane = tid & 63, wave = tid >> 6;
    const int Ho = H >> 1, Wo = W >> 1, lw = 31 - __builtin_clz(Wo), lh = 31 - __builtin_clz(Ho);
    const int P = B * Ho * Wo;
    float p0[8], p1[8], p2[8], acc[8][9];
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        // BatchNorm backward of encoder block 0: derived here from the sums the input-gradient kernel left (no finalise launch)
        if (fuse.mode == BNF_BWD) bn_fused_channel(fuse, cg * 8 + c, blockIdx.x == 0 && slot == 0, p0[c], p1[c], p2[c]);
        else { p0[c] = gcoef[cg * 8 + c]; p1[c] = gcoef[32 + cg * 8 + c]; p2[c] = gcoef[64 + cg * 8 + c]; }
#pragma unroll
        for (int t = 0; t < 9; ++t) acc[c][t] = 0.f;
    }
    // A thread takes 4 consecutive output pixels of one row per trip (as conv1_fwd_kernel): their 3x9 input window is two aligned
    // float4 loads + one scalar per row - 9 load instructions instead of 36 scalar ones (the loop was bound by the issue of its
    // vector-memory instructions, not by bytes) - and the 4 pixels' dz / y chunks are 8 more 16-byte loads, all in flight together.
    const int Wq = Wo >> 2, lq = lw - 2, Q = B * Ho * Wq;
    for (int q = blockIdx.x * 64 + slot; q < Q; q += gridDim.x * 64) {
        const int oxq = q & (Wq - 1), oy = (q >> lq) & (Ho - 1), b = q >> (lq + lh);
        const int ix0 = 8 * oxq;
        float win[3][9], dv[4][8], yv[4][8];
#pragma unroll
        for (int rr = 0; rr < 3; ++rr) {
            const int iy = 2 * oy + rr - 1;
            const bool rin = iy >= 0;
            const float* row = x + ((size_t)b * H + (rin ? iy : 0)) * W + ix0;
            const f32x4 v0 = *reinterpret_cast<const f32x4*>(row), v1 = *reinterpret_cast<const f32x4*>(row + 4);
            const float left = (ix0 > 0) ? row[-1] : 0.f;
            win[rr][0] = rin ? left : 0.f;
#pragma unroll
            for (int e = 0; e < 4; ++e) { win[rr][1 + e] = rin ? v0[e] : 0.f; win[rr][5 + e] = rin ? v1[e] : 0.f; }
        }
        const size_t p0i = ((size_t)b * Ho + oy) * Wo + 4 * oxq;
#pragma unroll
        for (int u = 0; u < 4; ++u) { load8<T>(dz + (p0i + u) * 32 + cg * 8, dv[u]); load8<T>(y + (p0i + u) * 32 + cg * 8, yv[u]); }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const float g = dv[u][c] * p0[c] + yv[u][c] * p1[c] + p2[c];
#pragma unroll
                for (int t = 0; t < 9; ++t) acc[c][t] += g * win[t / 3][2 * u + t % 3];
            }
        }
    }
#pragma unroll
    for (int c = 0; c < 8; ++c)
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const float s = cg_sum(acc[c][t]);
            if (lane < 4) red[wave][lane][c * 9 + t] = s;
        }
    __syncthreads();
    for (int j = tid; j < 288; j += 256) {
        const int t = j / 32, co = j % 32;  // slab layout [t][co] (CA=32, CB=1)
        float s = 0.f;
#pragma unroll
        for (int wv = 0; wv < 4; ++wv) s += red[wv][co >> 3][(co & 7) * 9 + t];
        slab[(size_t)blockIdx.x * 288 + j] = s;
    }
}

// ---------------------------------------------------------------------------
// final_layer tail: BN+LeakyReLU (on load) -> Conv2d(32->1,k3,s1,p1) -> Sigmoid  (models.py:78-81)
// fused with the reconstruction term of the ELBO (models.py:208) and its gradient:
//   xhat = sigmoid(logit);  bce += -(t*max(log xhat,-100) + (1-t)*max(log(1-xhat),-100))
//   dlogit = (xhat-t)/max(xhat(1-xhat),1e-12) * xhat(1-xhat) / N          (ATen BCE + sigmoid backward)
// Tile: 32x16 outputs per workgroup; phase 1 computes per input pixel the nine
// 32-channel dot products, phase 2 gathers the 3x3 neighbourhood from LDS.
struct ConvOutArgs {
    const void* yf; const float* coef;      // [B,H,W,32] T, rows sc,0,sh
    const float* wt; const float* bias;     // wt [9][32] (tap-major copy of final_layer.3.weight)
    const float* target;                    // [B,H,W] (= the input x)
    float* xhat; float* dlogit;             // [B,H,W]
    double* accum;                          // [0] bce sum
    int B, H, W; float inv_n; float slope;
};

template <typename T>
__global__ __launch_bounds__(256) void convout_fwd_kernel(ConvOutArgs a) {
    constexpr int TH = 16, TW = 32, PH = TH + 2, PW = TW + 2, NP = PH * PW;
    __shared__ float part[NP * 9];
    __shared__ float wred[4];
    const int tid = threadIdx.x;
    const int tiles_x = a.W / TW, tiles_y = a.H / TH;
    const int tile = blockIdx.x, tx = tile % tiles_x, ty = (tile / tiles_x) % tiles_y, b = tile / (tiles_x * tiles_y);
    const int y0 = ty * TH, x0 = tx * TW;
    const T* yf = reinterpret_cast<const T*>(a.yf);
    for (int pix = tid; pix < NP; pix += 256) {
        const int py = pix / PW, px = pix - py * PW, gy = y0 + py - 1, gx = x0 + px - 1;
        float acc[9];
#pragma unroll
        for (int t = 0; t < 9; ++t) acc[t] = 0.f;
        if (gy >= 0 && gy < a.H && gx >= 0 && gx < a.W) {
            const T* src = yf + (((size_t)b * a.H + gy) * a.W + gx) * 32;
#pragma unroll
            for (int cgp = 0; cgp < 4; ++cgp) {
                float v[8];
                load8<T>(src + cgp * 8, v);
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    const int ch = cgp * 8 + c;
                    const float av = leaky(v[c] * a.coef[ch] + a.coef[64 + ch], a.slope);
#pragma unroll
                    for (int t = 0; t < 9; ++t) acc[t] += av * a.wt[t * 32 + ch];
                }
            }
        }
#pragma unroll
        for (int t = 0; t < 9; ++t) part[pix * 9 + t] = acc[t];
    }
    __syncthreads();
    float bsum = 0.f;
    const float bo = a.bias[0];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int o = tid + k * 256, oy = o / TW, ox = o % TW;
        float logit = bo;
#pragma unroll
        for (int t = 0; t < 9; ++t) logit += part[((oy + t / 3) * PW + ox + t % 3) * 9 + t];
        const size_t gi = ((size_t)b * a.H + y0 + oy) * a.W + x0 + ox;
        const float tg = a.target[gi];
        const float xh = 1.f / (1.f + expf(-logit));
        const float l1 = fmaxf(logf(xh), -100.f), l0 = fmaxf(logf(1.f - xh), -100.f);
        bsum += -(tg * l1 + (1.f - tg) * l0);
        const float om = xh * (1.f - xh);
        a.xhat[gi] = xh;
        a.dlogit[gi] = (xh - tg) / fmaxf(om, 1e-12f) * om * a.inv_n;
    }
    bsum = wave_sum(bsum);
    if ((tid & 63) == 0) wred[tid >> 6] = bsum;
    __syncthreads();
    if (tid == 0) unsafeAtomicAdd(&a.accum[stat_rep() * 8 + 0], (double)(wred[0] + wred[1] + wred[2] + wred[3]));
}

// dlogit = g_xhat * xhat*(1-xhat) [+ gscale * dlogit_std]: caller-supplied dL/dxhat, optionally on top
// of the fused standard-ELBO gradient
static __global__ void dlogit_combine_kernel(const float* __restrict__ g, const float* __restrict__ xhat,
                                      const float* __restrict__ dstd, const float* __restrict__ gscale,
                                      float* __restrict__ dlogit, long n) {
    const float gs = gscale ? gscale[0] : 1.f;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        float v = 0.f;
        if (g) { const float xh = xhat[i]; v = g[i] * xh * (1.f - xh); }
        if (dstd) v += gs * dstd[i];
        dlogit[i] = v;
    }
}

// generic F.binary_cross_entropy(mean) forward + grad w.r.t. the prediction
static __global__ void bce_kernel(const float* __restrict__ xh_, const float* __restrict__ tg_, float* __restrict__ gx,
                           double* __restrict__ accum, long n, float inv_n) {
    __shared__ float wred[4];
    float bsum = 0.f;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const float xh = xh_[i], tg = tg_[i];
        bsum += -(tg * fmaxf(logf(xh), -100.f) + (1.f - tg) * fmaxf(logf(1.f - xh), -100.f));
        if (gx) gx[i] = (xh - tg) / fmaxf(xh * (1.f - xh), 1e-12f) * inv_n;
    }
    bsum = wave_sum(bsum);
    if ((threadIdx.x & 63) == 0) wred[threadIdx.x >> 6] = bsum;
    __syncthreads();
    if (threadIdx.x == 0) unsafeAtomicAdd(&accum[0], (double)(wred[0] + wred[1] + wred[2] + wred[3]));
}

// Backward of the output conv fused with the BN/LeakyReLU backward prologue of final_layer:
//   dA[p][c] = sum_t w[t][c] dl[p - off(t)];  dz = dA * leaky'(z);  dW[c][t] += a[p][c] * dl[p - off(t)]
//   stats: sum dz, sum dz*xhat_bn per channel; sum dl -> bias gradient.
struct ConvOutBwdArgs {
    const void* yf; const float* ocoef;   // layer block rows (stride 32)
    const float* wt; const float* dlogit; const float* gscale;
    void* dz; float* slab;                // slab [nWG][288] in [t][c] order
    double* stat;                         // [2][32]
    double* dbias;                        // sum of dlogit
    int B, H, W; float slope;
    float gmul;                           // gradient scale entering the backward (f16 storage; 1 otherwise)
};

template <typename T>
__global__ __launch_bounds__(256) void convout_bwd_kernel(ConvOutBwdArgs a) {
    __shared__ float red[4][4][89];
    const int tid = threadIdx.x, cg = tid & 3, slot = tid >> 2, lane = tid & 63, wave = tid >> 6;
    const int P = a.B * a.H * a.W, lw = 31 - __builtin_clz(a.W), lh = 31 - __builtin_clz(a.H);
    const T* yf = reinterpret_cast<const T*>(a.yf);
    T* dzp = reinterpret_cast<T*>(a.dz);
    const float gs = (a.gscale ? a.gscale[0] : 1.f) * a.gmul;
    float sc[8], sh[8], is[8], xm[8], wr[8][9], dw[8][9], s1[8], s2[8], sdl = 0.f;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const int ch = cg * 8 + c;
        sc[c] = a.ocoef[LC_SC * 32 + ch]; sh[c] = a.ocoef[LC_SH * 32 + ch];
        is[c] = a.ocoef[LC_INVSTD * 32 + ch]; xm[c] = a.ocoef[LC_XM * 32 + ch];
        s1[c] = 0.f; s2[c] = 0.f;
#pragma unroll
        for (int t = 0; t < 9; ++t) { wr[c][t] = a.wt[t * 32 + ch]; dw[c][t] = 0.f; }
    }
    for (int p = blockIdx.x * 64 + slot; p < P; p += gridDim.x * 64) {
        const int x = p & (a.W - 1), y = (p >> lw) & (a.H - 1); const size_t b = p >> (lw + lh);
        float dl[9], yv[8], o[8];
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int qy = y - (t / 3) + 1, qx = x - (t % 3) + 1;
            dl[t] = (qy >= 0 && qy < a.H && qx >= 0 && qx < a.W) ? a.dlogit[(b * a.H + qy) * a.W + qx] * gs : 0.f;
        }
        if (cg == 0) sdl += dl[4];
        load8<T>(yf + (size_t)p * 32 + cg * 8, yv);
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const float z = yv[c] * sc[c] + sh[c];
            const float av = leaky(z, a.slope);
            float da = 0.f;
#pragma unroll
            for (int t = 0; t < 9; ++t) { da += wr[c][t] * dl[t]; dw[c][t] += av * dl[t]; }
            const float dzv = round_as<T>(z > 0.f ? da : da * a.slope);
            o[c] = dzv; s1[c] += dzv; s2[c] += dzv * (yv[c] * is[c] + xm[c]);
        }
        store8<T>(dzp + (size_t)p * 32 + cg * 8, o);
    }
#pragma unroll
    for (int c = 0; c < 8; ++c) {
#pragma unroll
        for (int t = 0; t < 9; ++t) { const float s = cg_sum(dw[c][t]); if (lane < 4) red[wave][lane][c * 9 + t] = s; }
        const float a1 = cg_sum(s1[c]), a2 = cg_sum(s2[c]);
        if (lane < 4) { red[wave][lane][72 + c] = a1; red[wave][lane][80 + c] = a2; }
    }
    sdl = cg_sum(sdl);
    if (lane == 0) red[wave][0][88] = sdl;
    __syncthreads();
    for (int j = tid; j < 288; j += 256) {
        const int t = j / 32, ch = j % 32;
        float s = 0.f;
#pragma unroll
        for (int wv = 0; wv < 4; ++wv) s += red[wv][ch >> 3][(ch & 7) * 9 + t];
        a.slab[(size_t)blockIdx.x * 288 + j] = s;
    }
    if (tid < 64) {
        const int ch = tid & 31, k = tid >> 5;
        float s = 0.f;
#pragma unroll
        for (int wv = 0; wv < 4; ++wv) s += red[wv][ch >> 3][72 + k * 8 + (ch & 7)];
        unsafeAtomicAdd(&a.stat[stat_rep() * 64 + k * 32 + ch], (double)s);
    }
    if (tid == 64) unsafeAtomicAdd(a.dbias + stat_rep() * 8, (double)(red[0][0][88] + red[1][0][88] + red[2][0][88] + red[3][0][88]));
}

// ---------------------------------------------------------------------------
// BatchNorm: train-mode finalisation lives in common.cuh (BnFuse, folded into the consumer kernels).
// eval-mode coefficients from running statistics (evaluation path / model.eval())
static __global__ void bn_eval_coef_kernel(const float* gamma, const float* beta, const float* rm, const float* rv,
                                    float* block, int C, float eps) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const double invstd = 1.0 / sqrt((double)rv[c] + (double)eps);
    const double sc = (double)gamma[c] * invstd;
    block[LC_SC * C + c] = (float)sc; block[LC_ZERO * C + c] = 0.f;
    block[LC_SH * C + c] = (float)((double)beta[c] - (double)rm[c] * sc);
    block[LC_INVSTD * C + c] = (float)invstd; block[LC_XM * C + c] = (float)(-(double)rm[c] * invstd);
    block[LC_MEAN * C + c] = rm[c]; block[LC_VAR * C + c] = rv[c];
}
// ---------------------------------------------------------------------------
// latent block.  fc_mu | fc_var partial products arrive as split-K slabs.
struct LatentFwdArgs {
    const float* slab; int nslab, npad;             // [nslab][B][npad]
    const float* bmu; const float* bvar; const float* eps;
    float* mu; float* lv; float* z; double* accum;  // accum[1] += sum(1 + lv - mu^2 - exp(lv))
    int B, L;
};
constexpr int LAT_LANES = 32;   // lanes per (b,l): the split-K slabs are summed in parallel, all loads of a lane in flight
static __global__ void latent_fwd_kernel(LatentFwdArgs a) {
    __shared__ float wred[4];
    const int i = (blockIdx.x * blockDim.x + threadIdx.x) / LAT_LANES, sub = threadIdx.x & (LAT_LANES - 1);
    float term = 0.f;
    const bool ok = i < a.B * a.L;
    const int b = ok ? i / a.L : 0, l = ok ? i % a.L : 0;
    float m = 0.f, v = 0.f;
    if (ok) {
        const float* p = a.slab + (size_t)b * a.npad + l;
        const size_t ss = (size_t)a.B * a.npad;
#pragma unroll 4
        for (int s = sub; s < a.nslab; s += LAT_LANES) { m += p[s * ss]; v += p[s * ss + a.L]; }
    }
#pragma unroll
    for (int o = 1; o < LAT_LANES; o <<= 1) { m += __shfl_xor(m, o, 64); v += __shfl_xor(v, o, 64); }
    if (ok && sub == 0) {
        m += a.bmu[l]; v += a.bvar[l];
        const float sd = expf(0.5f * v);                      // models.py:181
        a.mu[i] = m; a.lv[i] = v; a.z[i] = a.eps[i] * sd + m;  // models.py:183
        term = 1.f + v - m * m - expf(v);                     // models.py:214
    }
    term = wave_sum(term);
    if ((threadIdx.x & 63) == 0) wred[threadIdx.x >> 6] = term;
    __syncthreads();
    if (threadIdx.x == 0) unsafeAtomicAdd(&a.accum[(blockIdx.x & (STAT_R - 1)) * 8 + 1], (double)(wred[0] + wred[1] + wred[2] + wred[3]));
}

// ELBO scalars (models.py:216-225): loss = bce + kld_weight*kld ; kld_loss reported with flipped sign.
// nrep: replicas of the accumulator block (STAT_R for a context's accumulators, 1 for the generic-loss buffer)
static __global__ void loss_finalize_kernel(const double* accum, float* out3, double inv_n, double inv_b, float kld_weight, int nrep) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        double a0 = 0.0, a1 = 0.0;
        for (int rep = 0; rep < nrep; ++rep) { a0 += accum[rep * 8 + 0]; a1 += accum[rep * 8 + 1]; }
        const double bce = a0 * inv_n, kld = -0.5 * a1 * inv_b;
        out3[0] = (float)(bce + (double)kld_weight * kld); out3[1] = (float)bce; out3[2] = (float)(-kld);
    }
}
static __global__ void kld_only_kernel(const float* mu, const float* lv, double* accum, int n, float k, float* gmu, float* glv) {
    __shared__ float wred[4];
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    float term = 0.f;
    if (i < n) {
        const float m = mu[i], v = lv[i], ev = expf(v);
        term = 1.f + v - m * m - ev;
        if (gmu) gmu[i] = k * m;                  // d(kld_weight*KL)/dmu
        if (glv) glv[i] = k * 0.5f * (ev - 1.f);  // d(kld_weight*KL)/dlog_var
    }
    term = wave_sum(term);
    if ((threadIdx.x & 63) == 0) wred[threadIdx.x >> 6] = term;
    __syncthreads();
    if (threadIdx.x == 0) unsafeAtomicAdd(&accum[1], (double)(wred[0] + wred[1] + wred[2] + wred[3]));
}

struct LatentBwdArgs {
    const float* slab; int nslab, npad;   // dz_lat partials [nslab][B][npad]
    const float* mu; const float* lv; const float* eps; const float* gscale;
    const float* gmu; const float* glv; const float* gz;   // optional external grads [B,L]
    float* dlat;                          // [B][2L]: dmu | dlv
    int B, L; float kld_weight; int add_kl;
    float gmul;                           // every upstream gradient entering here is multiplied by it (f16 gradient scaling; 1 otherwise)
};
static __global__ void latent_bwd_kernel(LatentBwdArgs a) {   // LAT_LANES lanes per (b,l)
    const int i = (blockIdx.x * blockDim.x + threadIdx.x) / LAT_LANES, sub = threadIdx.x & (LAT_LANES - 1);
    const bool ok = i < a.B * a.L;
    const int b = ok ? i / a.L : 0, l = ok ? i % a.L : 0;
    float d = 0.f;
    if (ok) {
        const float* p = a.slab + (size_t)b * a.npad + l;
        const size_t ss = (size_t)a.B * a.npad;
#pragma unroll 4
        for (int s = sub; s < a.nslab; s += LAT_LANES) d += p[s * ss];
    }
#pragma unroll
    for (int o = 1; o < LAT_LANES; o <<= 1) d += __shfl_xor(d, o, 64);
    if (!ok || sub != 0) return;
    if (a.gz) d += a.gz[i] * a.gmul;
    const float gs = (a.gscale ? a.gscale[0] : 1.f) * a.gmul;
    const float m = a.mu[i], v = a.lv[i], sd = expf(0.5f * v);
    float dmu = d, dlv = d * a.eps[i] * sd * 0.5f;
    if (a.add_kl) {
        const float k = gs * a.kld_weight / (float)a.B;
        dmu += k * m; dlv += k * 0.5f * (expf(v) - 1.f);
    }
    if (a.gmu) dmu += a.gmu[i] * a.gmul;
    if (a.glv) dlv += a.glv[i] * a.gmul;
    a.dlat[(size_t)b * 2 * a.L + l] = dmu; a.dlat[(size_t)b * 2 * a.L + a.L + l] = dlv;
}
// column sums of dlat -> fc_mu.bias / fc_var.bias gradients; one wave per column
static __global__ void colsum_kernel(const float* __restrict__ m, int rows, int cols, float* __restrict__ o0, float* __restrict__ o1, int split, float scale) {
    const int j = blockIdx.x;
    float s = 0.f;
    for (int r = threadIdx.x; r < rows; r += 64) s += m[(size_t)r * cols + j];
    s = wave_sum(s) * scale;
    if (threadIdx.x == 0) { if (j < split) o0[j] = s; else o1[j - split] = s; }
}

// f' (NHWC flatten: pix*256 + c) -> reference flatten index c*s2 + pix (models.py:133)
__device__ __forceinline__ int fref_of(int fp, int s2) { return (fp & 255) * s2 + (fp >> 8); }

// Block `blk` of 256 flatten positions, enumerated so that consecutive threads own consecutive
// REFERENCE indices (c*s2 + pix): stores of [.,F_ref]-major gradients coalesce; the strided reads
// of the small NHWC bottleneck tensor are absorbed by L2.
__device__ __forceinline__ void fmap_ref_major(int blk, int tid, int s2, int& fp, int& fr) {
    const int P = s2 < 16 ? s2 : 16, CW = 256 / P;          // pixels x channels per block (32-B read runs, 64-B write runs)
    const int pb = s2 / P;                                   // pixel blocks per channel group
    const int cblk = blk / pb, pblk = blk - cblk * pb;
    const int pix = pblk * P + (tid % P), c = cblk * CW + tid / P;
    fp = pix * 256 + c; fr = c * s2 + pix;
}

// pre_latents in the reference's NCHW-flatten order (types_helpers.py:20), on request only.
template <typename T>
__global__ void pre_latents_kernel(const T* __restrict__ y, const float* __restrict__ coef, float slope,
                                   float* __restrict__ out, int B, int F, int s2) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)B * F) return;
    const int fp = i % F; const long b = i / F; const int c = fp & 255;
    out[b * F + fref_of(fp, s2)] = leaky(tofloat(y[i]) * coef[c] + coef[512 + c], slope);
}

// da4 = dlat @ [Wmu;Wvar], then LeakyReLU/BN prologue of encoder block 3 backward.
template <typename T> struct FcDgradArgs {
    const float* dlat; const T* wp; int npad;   // dlat [B][2L]; wp packed [F/8][npad][8]
    const T* y; const float* ocoef; float slope;
    const float* gpre;                          // optional external grad on pre_latents [B,F] (reference order)
    T* dz; double* stat; int B, F, L2, s2;
    int bt_per_wg;   // batch rows per workgroup (multiple of 16)
    float gmul;      // scale of gpre (f16 gradient scaling; 1 otherwise)
};
template <typename T>
__global__ __launch_bounds__(256) void fc_dgrad_kernel(FcDgradArgs<T> a) {
    constexpr int BT = 16;
    extern __shared__ __attribute__((aligned(16))) float dl_s[];  // [L2][BT]
    const int tid = threadIdx.x, fp = blockIdx.x * 256 + tid;
    const int c = fp & 255;
    const float sc = a.ocoef[LC_SC * 256 + c], sh = a.ocoef[LC_SH * 256 + c];
    const float is = a.ocoef[LC_INVSTD * 256 + c], xm = a.ocoef[LC_XM * 256 + c];
    const T* wrow = a.wp + ((size_t)(fp >> 3) * a.npad) * 8 + (fp & 7);
    float s1 = 0.f, s2 = 0.f;
    const int bend = min(a.B, (int)(blockIdx.y + 1) * a.bt_per_wg);
    for (int b0 = blockIdx.y * a.bt_per_wg; b0 < bend; b0 += BT) {
        __syncthreads();
        for (int i = tid; i < a.L2 * BT; i += 256) {
            const int j = i / BT, bb = i % BT;
            dl_s[i] = (b0 + bb < a.B) ? a.dlat[(size_t)(b0 + bb) * a.L2 + j] : 0.f;
        }
        __syncthreads();
        float yv[BT], acc[BT];
#pragma unroll
        for (int bb = 0; bb < BT; ++bb) {   // y rows in flight while the dot products run
            acc[bb] = 0.f;
            yv[bb] = (b0 + bb < a.B) ? tofloat(a.y[(size_t)(b0 + bb) * a.F + fp]) : 0.f;
        }
        for (int j = 0; j < a.L2; ++j) {
            const float w = tofloat(wrow[(size_t)j * 8]);
#pragma unroll
            for (int q = 0; q < BT / 4; ++q) {
                const f32x4 d = *reinterpret_cast<const f32x4*>(&dl_s[j * BT + q * 4]);
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[q * 4 + e] += d[e] * w;
            }
        }
#pragma unroll
        for (int bb = 0; bb < BT; ++bb) {
            const int b = b0 + bb;
            if (b < a.B) {
                const size_t idx = (size_t)b * a.F + fp;
                float da = acc[bb];
                if (a.gpre) da += a.gpre[(size_t)b * a.F + fref_of(fp, a.s2)] * a.gmul;
                const float z = yv[bb] * sc + sh;
                const float dzv = round_as<T>(z > 0.f ? da : da * a.slope);
                a.dz[idx] = fromfloat<T>(dzv);
                s1 += dzv; s2 += dzv * (yv[bb] * is + xm);
            }
        }
    }
    unsafeAtomicAdd(&a.stat[stat_rep() * 512 + c], (double)s1);
    unsafeAtomicAdd(&a.stat[stat_rep() * 512 + 256 + c], (double)s2);
}

// The same product for 16-bit storage with 16-byte accesses: a thread owns 8 consecutive channels (one 16-byte group of the packed
// weight image, of y and of dz) and 4 batch rows, a workgroup 256 channels x 32 rows.  fc_dgrad_kernel moves two bytes per lane and
// memory instruction (64 of them per thread and 16 rows: it is bound by their issue, ~20 us for 16 MB); here 8 rows x channels cost
// one instruction.  Per element the arithmetic is fc_dgrad_kernel's (the same fma chain over j, the same epilogue expressions), so
// the stored dz is bit-identical; the statistics are summed in a different order.
template <typename T, int RT, int NRG>
__global__ __launch_bounds__(32 * NRG) void fc_dgrad8_kernel(FcDgradArgs<T> a) {
    static_assert(sizeof(T) == 2, "16-bit storage");
    static_assert(RT == 2 || RT == 4, "rows per thread");
    typedef typename H16<T>::v8 T8;
    constexpr int RB = NRG * RT, NTH = 32 * NRG;                     // rows per workgroup pass (NRG row groups x RT rows per thread)
    extern __shared__ __attribute__((aligned(16))) float dl8_s[];    // [L2][RB], then red[NRG][256][2]
    float* red = dl8_s + a.L2 * RB;
    const int tid = threadIdx.x, g = tid & 31, rg = tid >> 5, c0 = g * 8, f0 = blockIdx.x * 256 + c0;
    float sc[8], sh[8], is[8], xm[8], s1[8], s2[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        sc[e] = a.ocoef[LC_SC * 256 + c0 + e]; sh[e] = a.ocoef[LC_SH * 256 + c0 + e];
        is[e] = a.ocoef[LC_INVSTD * 256 + c0 + e]; xm[e] = a.ocoef[LC_XM * 256 + c0 + e];
        s1[e] = 0.f; s2[e] = 0.f;
    }
    const T* wrow = a.wp + (size_t)(f0 >> 3) * a.npad * 8;
    const int bend = min(a.B, (int)(blockIdx.y + 1) * a.bt_per_wg);
    for (int b0 = blockIdx.y * a.bt_per_wg; b0 < bend; b0 += RB) {
        __syncthreads();
        for (int i = tid; i < a.L2 * RB; i += NTH) {
            const int bb = i / a.L2, j = i - bb * a.L2;              // (consecutive threads read consecutive floats of a dlat row)
            dl8_s[j * RB + bb] = (b0 + bb < bend) ? a.dlat[(size_t)(b0 + bb) * a.L2 + j] : 0.f;
        }
        __syncthreads();
        T8 yv[RT];
        float acc[RT][8];
#pragma unroll
        for (int r = 0; r < RT; ++r) {
            const int b = b0 + rg * RT + r;
            yv[r] = *reinterpret_cast<const T8*>(a.y + (size_t)(b < bend ? b : b0) * a.F + f0);
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[r][e] = 0.f;
        }
        // eight weight groups in flight per block of the (sequential) j chain: one load per j costs the cache latency per j
        for (int j0 = 0; j0 < a.L2; j0 += 8) {
            T8 w[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) w[u] = *reinterpret_cast<const T8*>(wrow + (size_t)min(j0 + u, a.L2 - 1) * 8);
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                if (j0 + u < a.L2) {
                    float d[RT];
#pragma unroll
                    for (int r = 0; r < RT; ++r) d[r] = dl8_s[(j0 + u) * RB + rg * RT + r];
#pragma unroll
                    for (int r = 0; r < RT; ++r)
#pragma unroll
                        for (int e = 0; e < 8; ++e) acc[r][e] += d[r] * tofloat((T)w[u][e]);
                }
            }
        }
#pragma unroll
        for (int r = 0; r < RT; ++r) {
            const int b = b0 + rg * RT + r;
            if (b < bend) {
                T8 o;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    float da = acc[r][e];
                    if (a.gpre) da += a.gpre[(size_t)b * a.F + fref_of(f0 + e, a.s2)] * a.gmul;
                    const float y = tofloat((T)yv[r][e]);
                    const float z = y * sc[e] + sh[e];
                    const float dzv = round_as<T>(z > 0.f ? da : da * a.slope);
                    o[e] = fromfloat<T>(dzv);
                    s1[e] += dzv; s2[e] += dzv * (y * is[e] + xm[e]);
                }
                *reinterpret_cast<T8*>(a.dz + (size_t)b * a.F + f0) = o;
            }
        }
    }
    __syncthreads();
#pragma unroll
    for (int e = 0; e < 8; ++e) { red[(rg * 256 + c0 + e) * 2] = s1[e]; red[(rg * 256 + c0 + e) * 2 + 1] = s2[e]; }
    __syncthreads();
    if (tid < 256) {
        float v1 = 0.f, v2 = 0.f;
#pragma unroll
        for (int k = 0; k < NRG; ++k) { v1 += red[(k * 256 + tid) * 2]; v2 += red[(k * 256 + tid) * 2 + 1]; }
        unsafeAtomicAdd(&a.stat[stat_rep() * 512 + tid], (double)v1);
        unsafeAtomicAdd(&a.stat[stat_rep() * 512 + 256 + tid], (double)v2);
    }
}
static inline size_t fc_dgrad8_lds(int L2, int RT, int NRG) { return ((size_t)L2 * NRG * RT + NRG * 256 * 2) * 4; }

// dW_mu / dW_var [L][F_ref] = dlat^T @ a4  (K = batch)
template <typename T> struct FcWgradArgs {
    const float* dlat; const T* y; const float* coef; float slope;
    float* dwmu; float* dwvar; int B, F, L, s2;
    int bsplit;             // batch rows per grid.z slice; slice z writes slab z of dwmu / dwvar ([nz][L][F] each)
};
template <typename T>
__global__ __launch_bounds__(256) void fc_wgrad_kernel(FcWgradArgs<T> a) {
    constexpr int JT = 32, BC = 64;
    __shared__ __attribute__((aligned(16))) float dl_s[BC * JT];
    const int tid = threadIdx.x, j0 = blockIdx.y * JT, L2 = 2 * a.L;
    int fp, fr;
    fmap_ref_major(blockIdx.x, tid, a.s2, fp, fr);
    const int c = fp & 255;
    const float sc = a.coef[c], sh = a.coef[512 + c];
    float acc[JT];
#pragma unroll
    for (int j = 0; j < JT; ++j) acc[j] = 0.f;
    const int bz0 = blockIdx.z * a.bsplit, bz1 = min(a.B, bz0 + a.bsplit);
    for (int bc = bz0; bc < bz1; bc += BC) {
        __syncthreads();
        for (int i = tid; i < BC * JT; i += 256) {
            const int bb = i / JT, j = i % JT;
            dl_s[i] = (bc + bb < bz1 && j0 + j < L2) ? a.dlat[(size_t)(bc + bb) * L2 + j0 + j] : 0.f;
        }
        __syncthreads();
        const int nb = min(BC, bz1 - bc);
        for (int bb = 0; bb < nb; bb += 8) {   // 8 independent loads in flight per thread
            float av[8];
#pragma unroll
            for (int u = 0; u < 8; ++u)
                av[u] = (bb + u < nb) ? leaky(tofloat(a.y[(size_t)(bc + bb + u) * a.F + fp]) * sc + sh, a.slope) : 0.f;
#pragma unroll
            for (int u = 0; u < 8; ++u)
#pragma unroll
                for (int q = 0; q < JT / 4; ++q) {
                    const f32x4 d = *reinterpret_cast<const f32x4*>(&dl_s[(bb + u) * JT + q * 4]);
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[q * 4 + e] += d[e] * av[u];
                }
        }
    }
#pragma unroll
    for (int j = 0; j < JT; ++j) {
        const int jj = j0 + j;
        const size_t zoff = (size_t)blockIdx.z * a.L * a.F;
        if (jj < a.L) a.dwmu[zoff + (size_t)jj * a.F + fr] = acc[j];
        else if (jj < L2) a.dwvar[zoff + (size_t)(jj - a.L) * a.F + fr] = acc[j];
    }
}

// decoder_input forward (models.py:162): d0[b][f'] = bd[f] + sum_l z[b][l] Wd[f][l]
template <typename T>
__global__ __launch_bounds__(256) void decin_fwd_kernel(const float* __restrict__ z, const float* __restrict__ wd,
                                                        const float* __restrict__ bd, T* __restrict__ d0, int B, int F, int L, int s2) {
    constexpr int BT = 16;
    extern __shared__ __attribute__((aligned(16))) float z_s[];  // [BT][L]
    const int tid = threadIdx.x, fp = blockIdx.x * 256 + tid, b0 = blockIdx.y * BT, fr = fref_of(fp, s2);
    for (int i = tid; i < BT * L; i += 256) z_s[i] = (b0 + i / L < B) ? z[(size_t)b0 * L + i] : 0.f;
    __syncthreads();
    float acc[BT];
    const float bias = bd[fr];
#pragma unroll
    for (int bb = 0; bb < BT; ++bb) acc[bb] = bias;
    const float* wrow = wd + (size_t)fr * L;
    if ((L & 3) == 0) {
        for (int l = 0; l < L; l += 4) {
            const f32x4 w = *reinterpret_cast<const f32x4*>(wrow + l);
#pragma unroll
            for (int bb = 0; bb < BT; ++bb) {
                const f32x4 zz = *reinterpret_cast<const f32x4*>(&z_s[bb * L + l]);
                acc[bb] += zz[0] * w[0] + zz[1] * w[1] + zz[2] * w[2] + zz[3] * w[3];
            }
        }
    } else {   // latent sizes that are not a multiple of 4 (the reference's default is 10): rows are not 16-byte aligned
        for (int l = 0; l < L; ++l) {
            const float w = wrow[l];
#pragma unroll
            for (int bb = 0; bb < BT; ++bb) acc[bb] += z_s[bb * L + l] * w;
        }
    }
#pragma unroll
    for (int bb = 0; bb < BT; ++bb)
        if (b0 + bb < B) d0[(size_t)(b0 + bb) * F + fp] = fromfloat<T>(acc[bb]);
}

// decoder_input weight/bias gradient: dWd[f][l] = sum_b dd0[b][f'] z[b][l]; dbd[f] = sum_b dd0[b][f']
template <typename T>
__global__ __launch_bounds__(256) void decin_wgrad_kernel(const T* __restrict__ dd0, const float* __restrict__ z,
                                                          float* __restrict__ dwd, float* __restrict__ dbd, int B, int F, int L, int s2,
                                                          int bsplit) {   // grid.z slice z -> slab z of dwd ([nz][F][L]) and dbd ([nz][F])
    constexpr int LT = 32, BC = 64;
    __shared__ __attribute__((aligned(16))) float z_s[BC * LT];
    const int tid = threadIdx.x, l0 = blockIdx.y * LT;
    int fp, fr;
    fmap_ref_major(blockIdx.x, tid, s2, fp, fr);
    float acc[LT], sb = 0.f;
#pragma unroll
    for (int l = 0; l < LT; ++l) acc[l] = 0.f;
    const int bz0 = blockIdx.z * bsplit, bz1 = min(B, bz0 + bsplit);
    for (int bc = bz0; bc < bz1; bc += BC) {
        __syncthreads();
        for (int i = tid; i < BC * LT; i += 256) {
            const int bb = i / LT, l = i % LT;
            z_s[i] = (bc + bb < bz1 && l0 + l < L) ? z[(size_t)(bc + bb) * L + l0 + l] : 0.f;
        }
        __syncthreads();
        const int nb = min(BC, bz1 - bc);
        for (int bb = 0; bb < nb; bb += 8) {   // 8 independent loads in flight per thread
            float gv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) gv[u] = (bb + u < nb) ? tofloat(dd0[(size_t)(bc + bb + u) * F + fp]) : 0.f;
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                sb += gv[u];
#pragma unroll
                for (int q = 0; q < LT / 4; ++q) {
                    const f32x4 zz = *reinterpret_cast<const f32x4*>(&z_s[(bb + u) * LT + q * 4]);
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[q * 4 + e] += zz[e] * gv[u];
                }
            }
        }
    }
#pragma unroll
    for (int l = 0; l < LT; ++l)
        if (l0 + l < L) dwd[(size_t)blockIdx.z * F * L + (size_t)fr * L + l0 + l] = acc[l];
    if (blockIdx.y == 0) dbd[(size_t)blockIdx.z * F + fr] = sb;
}

// ---------------------------------------------------------------------------
// weight packing: f32 reference layouts -> MFMA B-operand images [tap][K/8][N][8] of T
template <typename T>
__global__ void pack_kernel(const PackDesc* __restrict__ descs) {
    const PackDesc d = descs[blockIdx.y];
    T* dst = reinterpret_cast<T*>(d.dst);
    // 32-bit index arithmetic (every image is far below 2^31 elements): the 64-bit divisions dominated this kernel
    const int total = (int)d.n;
    if (d.kind == 0) {
        // conv weights [A][Bc][9] -> [tap][K/8][N][8]: a thread reads the 9 contiguous taps of one (a, b) pair (coalesced
        // 36-byte reads) and scatters them to the 9 tap planes
        const int K = d.k_is_first ? d.A : d.Bc, N = d.k_is_first ? d.Bc : d.A, pairs = d.A * d.Bc;
        for (int pr = blockIdx.x * blockDim.x + threadIdx.x; pr < pairs; pr += gridDim.x * blockDim.x) {
            const int ai = pr / d.Bc, bi = pr - ai * d.Bc;
            const int k = d.k_is_first ? ai : bi, n = d.k_is_first ? bi : ai;
            const float* sp = d.src + (size_t)pr * 9;
            float w[9];
#pragma unroll
            for (int t = 0; t < 9; ++t) w[t] = sp[t];
            const int o = ((k >> 3) * N + n) * 8 + (k & 7), plane = (K >> 3) * N * 8;
#pragma unroll
            for (int t = 0; t < 9; ++t) dst[t * plane + o] = fromfloat<T>(w[t]);
        }
        return;
    }
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        if (d.kind == 0) {
        } else if (d.kind == 1) {
            const int e = i & 7; const int r = i >> 3; const int n = r % d.npad; const int fp = (r / d.npad) * 8 + e;
            const int F = 256 * d.s2; const int fr = (fp & 255) * d.s2 + (fp >> 8);
            float v = 0.f;
            if (n < d.L) v = d.src[(long)n * F + fr]; else if (n < 2 * d.L) v = d.src2[(long)(n - d.L) * F + fr];
            dst[i] = fromfloat<T>(v);
        } else if (d.kind == 2) {
            const int e = i & 7; const int r = i >> 3; const int n = r % d.npad; const int fp = (r / d.npad) * 8 + e;
            const int fr = (fp & 255) * d.s2 + (fp >> 8);
            dst[i] = fromfloat<T>(n < d.L ? d.src[(long)fr * d.L + n] : 0.f);
        } else {
            const int t = i / d.A, c = i % d.A;   // dst f32 [9][C] from src [C][9]
            reinterpret_cast<float*>(d.dst)[i] = d.src[c * 9 + t];
        }
    }
}

// ---------------------------------------------------------------------------
// torch.optim.AdamW (train.py:228) over up to two contiguous parameter ranges
// (encoder group, decoder group), each with its own OneCycle lr / beta1 (train.py:233-238).
// every float below is a double expression of the host rounded once (vae_adamw_step): omb1 = 1 - beta1, decay = 1 - lr * weight_decay,
// step_size = lr / (1 - beta1^t), inv_sqrt_bc2 = 1 / sqrt(1 - beta2^t), omb2 = 1 - beta2
struct AdamGroup { long off, n; float beta1, omb1, decay; float step_size, inv_sqrt_bc2; };
struct AdamArgs {
    float* p; const float* g; float* m; float* v;
    AdamGroup grp[2]; int ngrp;
    float beta2, omb2, eps, grad_scale; int step;
};
// One element of torch.optim.AdamW's single-tensor update (decoupled decay, bias corrections as torch computes them).
__device__ __forceinline__ void adamw_one(float& p, float g, float& m, float& v, const AdamGroup& gr, const AdamArgs& a, float decay) {
    g *= a.grad_scale;
    p *= decay;
    m = m * gr.beta1 + gr.omb1 * g;
    v = v * a.beta2 + a.omb2 * g * g;
    const float denom = sqrtf(v) * gr.inv_sqrt_bc2 + a.eps;
    p -= gr.step_size * (m / denom);
}
// 16-byte accesses (the flat buffers' ranges start on 256-byte boundaries); a range whose length is not a multiple of four
// finishes with scalar elements.  The bias corrections arrive as kernel arguments: computed per thread (two f64 pow, a sqrt and a
// division each) they cost more than the update itself.
static __global__ void adamw_kernel(AdamArgs a) {
    const AdamGroup gr = a.grp[blockIdx.y];
    const float decay = gr.decay;
    const long n4 = (gr.off & 3) == 0 ? gr.n >> 2 : 0;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        const long k = gr.off + 4 * i;
        f32x4 p = *reinterpret_cast<const f32x4*>(a.p + k), m = *reinterpret_cast<const f32x4*>(a.m + k), v = *reinterpret_cast<const f32x4*>(a.v + k);
        const f32x4 g = *reinterpret_cast<const f32x4*>(a.g + k);
#pragma unroll
        for (int e = 0; e < 4; ++e) { float pe = p[e], me = m[e], ve = v[e]; adamw_one(pe, g[e], me, ve, gr, a, decay); p[e] = pe; m[e] = me; v[e] = ve; }
        *reinterpret_cast<f32x4*>(a.p + k) = p; *reinterpret_cast<f32x4*>(a.m + k) = m; *reinterpret_cast<f32x4*>(a.v + k) = v;
    }
    for (long i = 4 * n4 + (long)blockIdx.x * blockDim.x + threadIdx.x; i < gr.n; i += (long)gridDim.x * blockDim.x) {
        const long k = gr.off + i;
        float pe = a.p[k], me = a.m[k], ve = a.v[k];
        adamw_one(pe, a.g[k], me, ve, gr, a, decay);
        a.p[k] = pe; a.m[k] = me; a.v[k] = ve;
    }
}

// ---------------------------------------------------------------------------
// counter-based generator restated in oracle/vae_oracle.py (splitmix64 of seed, stream, counter)
__device__ __forceinline__ unsigned long long splitmix64(unsigned long long x) {
    x += 0x9E3779B97F4A7C15ULL;
    unsigned long long z = x;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}
__device__ __forceinline__ double counter_uniform(unsigned long long i, unsigned long long seed, unsigned long long stream) {
    unsigned long long base = splitmix64(seed);
    base = splitmix64(base ^ (stream * 0xD1342543DE82EF95ULL));
    const unsigned long long bits = splitmix64(base + i * 0x2545F4914F6CDD1DULL);
    return (double)(bits >> 11) * (1.0 / 9007199254740992.0);
}
// eps ~ N(0,1): Box-Muller on the counter generator (same as oracle.counter_normal(n, seed, 5))
static __global__ void counter_normal_kernel(float* out, long n, unsigned long long seed, unsigned long long stream) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double u1 = counter_uniform(i, seed, 2 * stream + 1000003ULL), u2 = counter_uniform(i, seed, 2 * stream + 1000004ULL);
    out[i] = (float)(sqrt(-2.0 * log(1.0 - u1)) * cos(2.0 * 3.14159265358979323846 * u2));
}
// o[0] = scale * sum over the STAT_R replicas of an accumulator slot
static __global__ void accum_to_f32_kernel(const double* slot, float* o, float scale) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        double s = 0.0;
        for (int rep = 0; rep < STAT_R; ++rep) s += slot[rep * 8];
        o[0] = (float)(s * (double)scale);
    }
}
template <typename T>
__global__ void nhwc_to_nchw_kernel(const T* in, float* out, long n, int C, int HW) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int ch = i % C; const long pix = (i / C) % HW; const long b = i / ((long)C * HW);
    out[(b * C + ch) * HW + pix] = tofloat(in[i]);
}
