// Shared device helpers for the gfx950 (MI355X / CDNA4) VAE-step kernels.
// Wavefront = 64 lanes; MFMA 32x32 tiles; all kernels use 256-thread workgroups.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __bf16 bf16;
typedef _Float16 f16;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;

// Phase-ablation bits of the timing diagnostics (tools/diag/gpu_ablate*.py): compiled in only by `make STAMPS=1`; in the product
// build the tests fold to constants and the branches disappear from the hot loops.
#ifdef VAE_PHASE_STAMPS
#define VAE_ABLATE(word, bits) (((word) & (bits)) != 0)
#else
#define VAE_ABLATE(word, bits) false
#endif

static constexpr int WG_KP = 64;  // weight-gradient kernels: low-res pixels per K tile
#ifndef WG_SPAD_V
#define WG_SPAD_V 64
#endif
#ifndef WG_GPAD_V
#define WG_GPAD_V 32
#endif
static constexpr int WG_SPAD = WG_SPAD_V, WG_GPAD = WG_GPAD_V;  // row pads (bytes) of the staged low-res / high-res operand

#define LDS_PTR(T) T __attribute__((address_space(3)))*

__device__ __forceinline__ float bf2f(bf16 v) { return (float)v; }
__device__ __forceinline__ float tofloat(float v) { return v; }
__device__ __forceinline__ float tofloat(bf16 v) { return (float)v; }
__device__ __forceinline__ float tofloat(f16 v) { return (float)v; }
template <typename T> __device__ __forceinline__ T fromfloat(float v);
template <> __device__ __forceinline__ float fromfloat<float>(float v) { return v; }
template <> __device__ __forceinline__ bf16 fromfloat<bf16>(float v) { return (bf16)v; }
template <> __device__ __forceinline__ f16 fromfloat<f16>(float v) { return (f16)v; }

// the two 16-bit storage types share every code path; H16<T> names their vector types
template <typename T> struct H16;
template <> struct H16<bf16> { typedef bf16x8 v8; typedef bf16x2 v2; };
template <> struct H16<f16> { typedef f16x8 v8; typedef f16x2 v2; };

// Round a value the way it will be stored (so statistics see the stored tensor).
template <typename T> __device__ __forceinline__ float round_as(float v) { return tofloat(fromfloat<T>(v)); }

// ---- 16-byte vector of storage elements -----------------------------------
template <typename T> struct Vec16;
template <> struct Vec16<float> {
    static constexpr int N = 4;
    f32x4 v;
    __device__ __forceinline__ float get(int i) const { return v[i]; }
    __device__ __forceinline__ void set(int i, float x) { v[i] = x; }
};
template <> struct Vec16<bf16> {
    static constexpr int N = 8;
    bf16x8 v;
    __device__ __forceinline__ float get(int i) const { return (float)v[i]; }
    __device__ __forceinline__ void set(int i, float x) { v[i] = (bf16)x; }
};
template <> struct Vec16<f16> {
    static constexpr int N = 8;
    f16x8 v;
    __device__ __forceinline__ float get(int i) const { return (float)v[i]; }
    __device__ __forceinline__ void set(int i, float x) { v[i] = (f16)x; }
};
template <typename T> __device__ __forceinline__ Vec16<T> zero_vec16() {
    Vec16<T> r;
#pragma unroll
    for (int i = 0; i < Vec16<T>::N; ++i) r.set(i, 0.f);
    return r;
}

// Streaming (read-once / write-once) accesses: non-temporal, so the activation streams do not evict the
// weight images that every workgroup re-reads from L2.
template <typename T> __device__ __forceinline__ Vec16<T> load_stream(const T* p) {
    Vec16<T> r;
    r.v = __builtin_nontemporal_load(reinterpret_cast<const decltype(r.v)*>(p));
    return r;
}
template <typename T> __device__ __forceinline__ void store_stream(T* p, const Vec16<T>& v) {
    __builtin_nontemporal_store(v.v, reinterpret_cast<decltype(v.v)*>(p));
}

// ---- MFMA fragment: 8 k-values per lane, k = 8*(lane>>5) + j ---------------
// bf16: one v_mfma_f32_32x32x16_bf16.  f32: eight v_mfma_f32_32x32x2_f32, step j
// contracting k in {j, 8+j}; the k permutation is the same for A and B, so the
// product is exact f32 FMA-chain arithmetic over the same 16 k values.
template <typename T> struct Frag;
template <> struct Frag<float> {
    float v[8];
    __device__ __forceinline__ void set(int j, float x) { v[j] = x; }
};
template <> struct Frag<bf16> {
    bf16x8 v;
    __device__ __forceinline__ void set(int j, float x) { v[j] = (bf16)x; }
};

template <> struct Frag<f16> {
    f16x8 v;
    __device__ __forceinline__ void set(int j, float x) { v[j] = (f16)x; }
};

__device__ __forceinline__ void mma(f32x16& acc, const Frag<float>& a, const Frag<float>& b) {
#pragma unroll
    for (int j = 0; j < 8; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.v[j], b.v[j], acc, 0, 0, 0);
}
__device__ __forceinline__ void mma(f32x16& acc, const Frag<bf16>& a, const Frag<bf16>& b) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.v, b.v, acc, 0, 0, 0);
}

__device__ __forceinline__ void mma(f32x16& acc, const Frag<f16>& a, const Frag<f16>& b) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a.v, b.v, acc, 0, 0, 0);
}

// 8 consecutive elements from a generic (global or LDS) pointer, 16-B aligned.
__device__ __forceinline__ Frag<float> load_frag(const float* p) {
    Frag<float> f;
    f32x4 a = *reinterpret_cast<const f32x4*>(p);
    f32x4 b = *reinterpret_cast<const f32x4*>(p + 4);
#pragma unroll
    for (int j = 0; j < 4; ++j) { f.v[j] = a[j]; f.v[4 + j] = b[j]; }
    return f;
}
__device__ __forceinline__ Frag<bf16> load_frag(const bf16* p) {
    Frag<bf16> f;
    f.v = *reinterpret_cast<const bf16x8*>(p);
    return f;
}

__device__ __forceinline__ Frag<f16> load_frag(const f16* p) {
    Frag<f16> f;
    f.v = *reinterpret_cast<const f16x8*>(p);
    return f;
}

// accumulator element i of a 32x32 tile sits at row acc_row(i, lane), col lane&31
__device__ __forceinline__ int acc_row(int i, int lane) { return (i & 3) + 8 * (i >> 2) + 4 * (lane >> 5); }

// ---- per-channel affine (+leaky) transform applied while staging a tensor --
//   v = leaky( t0 * p0[c] + t1 * p1[c] + p2[c], slope )
// activation load : t0 = raw conv output y, p = (gamma*invstd, 0, beta - mean*gamma*invstd), slope 0.01
// gradient load   : t0 = dz, t1 = y, p = BatchNorm-backward coefficients, slope 1
// LeakyReLU for 0 <= slope <= 1 (0.01 in the model, 1 = identity for gradient operands): max(z, slope*z) is the
// same value as the select form, in two instructions instead of three
__device__ __forceinline__ float leaky(float z, float slope) { return fmaxf(z, z * slope); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// Per-layer statistics / coefficient block, C floats per row:
//  0 sc   1 zero  2 sh   3 invstd  4 xm(=-mean*invstd)  5 p0  6 p1  7 p2  8 mean  9 var
enum { LC_SC = 0, LC_ZERO = 1, LC_SH = 2, LC_INVSTD = 3, LC_XM = 4, LC_P0 = 5, LC_P1 = 6, LC_P2 = 7, LC_MEAN = 8, LC_VAR = 9, LC_ROWS = 10 };

// ---- BatchNorm finalisation folded into the consumer kernel ------------------------------------------------
// A train-mode BatchNorm layer is "finalised" (batch statistics -> per-channel staging coefficients) by whichever
// kernel stages its tensor next: every workgroup derives the coefficients of the channels it needs in its
// prologue (a few f64 operations per channel), and one designated workgroup also writes the layer's coefficient
// block, running statistics (forward) or dgamma/dbeta (backward).  This removes one dependent launch per layer
// boundary.  Forward (models.py:46,69,78 of the reference: BatchNorm2d, momentum 0.1, unbiased running variance):
//   sc = gamma*invstd, sh = beta - mean*sc.   Backward: dL/dy = dz*p0 + y*p1 + p2 with p0 = gamma*invstd,
//   p1 = -p0*invstd*mean(dz*xhat), p2 = -p0*mean(dz) + p0*mean(dz*xhat)*mean*invstd.
// Statistics and scalar accumulators are kept in STAT_R replicas (a workgroup adds into replica blockIdx & (STAT_R-1)):
// thousands of workgroups finishing together otherwise serialise on the same few f64 atomics (~12 ns each, measured as a
// 25 us tail at 2048 workgroups).  Readers sum the replicas.  Layouts: stat[rep][2][C], accum[rep][8].
static constexpr int STAT_R = 8;
__device__ __forceinline__ double stat_sum(const double* stat, int C, int idx) {
    double s = 0.0;
#pragma unroll
    for (int rep = 0; rep < STAT_R; ++rep) s += stat[rep * 2 * C + idx];
    return s;
}
__device__ __forceinline__ int stat_rep() { return (blockIdx.x + blockIdx.y + blockIdx.z) & (STAT_R - 1); }

struct BnFuse {
    const double* stat;             // forward: [sum y | sum y^2]; backward: [sum dz | sum dz*xhat]   (2*C)
    const float* gamma; const float* beta;
    float* block;                   // LC_* rows, stride C
    float* running_mean; float* running_var; long long* nbt;
    float* dgamma; float* dbeta; float* dconv_bias;
    double count, inv_count; float eps, momentum;   // inv_count = 1 / count (host)
    float ginv;                     // backward: dgamma / dbeta are written times ginv (f16 gradient scaling, vae_ctx::ginv)
    int C, mode, update_running;    // mode 0: coefficients are read from the block; 1: forward; 2: backward
};
enum { BNF_NONE = 0, BNF_FWD = 1, BNF_BWD = 2 };

__device__ __forceinline__ void bn_fused_channel(const BnFuse& f, int c, bool writer, float& k0, float& k1, float& k2,
                                                 float* invstd_out = nullptr, float* xm_out = nullptr) {
    const int C = f.C;
    if (f.mode == BNF_FWD) {
        // (every workgroup of every consumer kernel runs this in its prologue: no f64 division or square root on that path - the
        //  reciprocal count comes from the host, 1/sqrt from v_rsq_f64 refined by two Newton steps to full double precision)
        const double mean = stat_sum(f.stat, C, c) * f.inv_count;
        double var = stat_sum(f.stat, C, C + c) * f.inv_count - mean * mean;
        if (var < 0) var = 0;
        const double xv = var + (double)f.eps;
        double invstd = __builtin_amdgcn_rsq(xv);
        invstd = invstd * (1.5 - 0.5 * xv * invstd * invstd);
        invstd = invstd * (1.5 - 0.5 * xv * invstd * invstd);
        const double sc = (double)f.gamma[c] * invstd;
        k0 = (float)sc; k1 = 0.f; k2 = (float)((double)f.beta[c] - mean * sc);
        if (invstd_out) { *invstd_out = (float)invstd; *xm_out = (float)(-mean * invstd); }   // (= block[LC_INVSTD], block[LC_XM])
        if (writer) {
            f.block[LC_SC * C + c] = k0; f.block[LC_ZERO * C + c] = 0.f; f.block[LC_SH * C + c] = k2;
            f.block[LC_INVSTD * C + c] = (float)invstd; f.block[LC_XM * C + c] = (float)(-mean * invstd);
            f.block[LC_MEAN * C + c] = (float)mean; f.block[LC_VAR * C + c] = (float)var;
            if (f.update_running) {
                const double unb = f.count > 1 ? var * f.count / (f.count - 1) : var;
                f.running_mean[c] = (float)((1.0 - f.momentum) * f.running_mean[c] + f.momentum * mean);
                f.running_var[c] = (float)((1.0 - f.momentum) * f.running_var[c] + f.momentum * unb);
                if (c == 0 && f.nbt) f.nbt[0] += 1;
            }
        }
    } else {
        const double sdz = stat_sum(f.stat, C, c), sdzx = stat_sum(f.stat, C, C + c);
        const double invstd = f.block[LC_INVSTD * C + c], mean = f.block[LC_MEAN * C + c];
        const double s = (double)f.gamma[c] * invstd, m1 = sdz * f.inv_count, m2 = sdzx * f.inv_count;
        k0 = (float)s; k1 = (float)(-s * m2 * invstd); k2 = (float)(-s * m1 + s * m2 * mean * invstd);
        if (writer) {
            f.block[LC_P0 * C + c] = k0; f.block[LC_P1 * C + c] = k1; f.block[LC_P2 * C + c] = k2;
            f.dgamma[c] = (float)(sdzx * (double)f.ginv); f.dbeta[c] = (float)(sdz * (double)f.ginv);
            if (f.dconv_bias) f.dconv_bias[c] = 0.f;
        }
    }
}
// standalone finalisation (consumers without the fused prologue, e.g. the one-tile-per-workgroup kernels)
static __global__ void bn_finalize_kernel(BnFuse f) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= f.C) return;
    float k0, k1, k2;
    bn_fused_channel(f, c, true, k0, k1, k2);
}

#define HIP_CHECK_RET(expr)                                                       \
    do {                                                                          \
        hipError_t _e = (expr);                                                   \
        if (_e != hipSuccess) return vae_set_error(#expr, hipGetErrorString(_e)); \
    } while (0)
