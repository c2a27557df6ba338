// Host-side context of the VAE step: shared by the C-ABI translation unit (vae_api.hip) and the per-storage-type
// translation units (impl_bf16.hip / impl_f16.hip / impl_f32.hip) that instantiate the templated launch code.
#pragma once
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>
#include <math.h>
#include <string>
#include <vector>
#include <algorithm>

#include "../../include/vae_step.h"

int vae_set_error(const char* what, const char* why);   // vae_api.hip; message readable through vae_last_error()

#include "common.cuh"

#define LAUNCH_CHECK(name)                                                        \
    do {                                                                          \
        hipError_t _e = hipGetLastError();                                        \
        if (_e != hipSuccess) return vae_set_error(name, hipGetErrorString(_e)); \
    } while (0)

static const int kBnC[8] = {32, 64, 128, 256, 128, 64, 32, 32};
static const float kSlope = 0.01f;   // nn.LeakyReLU() default (models.py:47,70,79)
static const float kBnEps = 1e-5f;   // nn.BatchNorm2d default eps
static const float kBnMom = 0.1f;    // nn.BatchNorm2d default momentum

static inline int64_t align_up(int64_t v, int64_t a) { return (v + a - 1) / a * a; }
static inline int ilog2(int v) { int l = 0; while ((1 << l) < v) ++l; return l; }

struct Tiling { int lth, ltw, lTB, tiles_x, tiles_y; };
static Tiling make_tiling(int Hs, int Ws, int pixels) {
    const int tw = std::min(Ws, pixels >= 128 ? 16 : 8), th = std::min(Hs, pixels / tw), TB = pixels / (th * tw);
    Tiling t; t.lth = ilog2(th); t.ltw = ilog2(tw); t.lTB = ilog2(TB); t.tiles_x = Ws / tw; t.tiles_y = Hs / th;
    return t;
}

struct BnLayer {
    int C, H, W;            // spatial size of the tensor this BN normalises
    double* stat_f; double* stat_b; float* block; void* y; void* dz;
    int p_gamma, p_beta, p_convw, p_convb;
    // deep layers, 16-bit storage: materialised LeakyReLU(BN(y)) / BatchNorm-backward gradient g (written as a side effect by the
    // pipelined kernel that stages the tensor first) for the weight-gradient kernels; *_ok: valid for the current forward / backward
    void* act = nullptr; void* dy = nullptr; int act_ok = 0, dy_ok = 0;
};

// weight packing descriptors (pack_kernel, edge_kernels.cuh): f32 reference layouts -> MFMA B-operand images of T
struct PackDesc {
    const float* src; const float* src2; void* dst;
    int kind;     // 0 conv [A][Bc][9]; 1 fc (mu|var -> [F/8][npad][8]); 2 decoder_input; 3 tap-major f32 copy [9][C]
    int A, Bc, k_is_first, npad, L, s2;
    long n;       // elements of dst
};

// split-K sizing of the weight-gradient kernels (vae_set_option knobs; slabs are sized at vae_create for the defaults).
// Workgroup targets: weight gradients run beside the input-gradient chain; on a saturated GPU (large batch x image) few
// workgroups keep them out of its way (-4 % step time at the bench workload), a small problem wants them everywhere.
//   wide: 128x32-channel tiles on 8 waves where the low-res side has >= 128 channels (16-bit prefetching kernel)
//   tile 1: 64x32 channel tiles (prefetching kernel) also where 64x64 would fit
struct WgradKnobs { int wgs = 128, cap_mb = 48, tile = 1, wide = 1, wide_wgs = 128, small_wgs = 1024, mid8 = 0, force_simple = 0; };

struct vae_ctx {
    int H, L, maxB, dtype, gen, s, s2; int64_t F; int npad_fc, npad_di; size_t esz;
    int64_t poff[VAE_NUM_PARAMS], psz[VAE_NUM_PARAMS], ptotal, bnoff[8], bnc[8], bntotal;
    BnLayer lay[8];
    void *d0, *dd0;
    float *eps, *dlat, *dlogit, *dlogit2, *ident, *wout_t;
    void* wp_fwd[8]; void* wp_dg[8];   // indexed by BN layer id (1..7); [0] unused
    void *fcpack, *dipack;
    PackDesc* d_descs; std::vector<PackDesc> h_descs; const float* packed_for;
    float* slab; size_t slab_floats;
    // side streams for work only the optimiser consumes (weight gradients, their split-K reductions) and for weight packing
    static constexpr int NSIDE = 3, NFORK = 16;
    hipStream_t side[NSIDE]; float* side_slab[NSIDE]; hipEvent_t ev_fork[NFORK], ev_join[NSIDE], ev_pack; int side_rr, fork_rr, n_side_ok;
    static constexpr int NBUCKET = 2; hipEvent_t ev_bucket[NBUCKET];   // bucketed gradient exchange (vae_train_step_fused): bucket i reduced on the communication stream
    hipStream_t comm; hipEvent_t ev_comm; int comm_busy;   // stream lent to the caller for the mid-backward gradient all-reduce (vae_comm_stream)
    void* nccl_comm = nullptr; int comm_rank = 0, comm_world = 0;   // RCCL communicator owned by the context (vae_comm.hip)
    int use_side_stream, knob_bwd_per_cu, knob_wave_nt_max, knob_lay22_min_nt, knob_down_waves, knob_pack_grid, knob_xcd_map, knob_up_nt_max, knob_lay42, knob_wgrad_layer_wgs, knob_conv1_grid, use_fused_bn, knob_rev, knob_lean, walk_dir, bwd_dirty, bwd_half_done;
    double* dstats; size_t n_dstats; double* accum;  // accum: [0] bce, [1] kl term, [2] sum dlogit
    double* generic_accum;                           // vae_elbo_generic on this context's device (per context, not process-global)
    WgradKnobs wk;
    float* reduce_tmp = nullptr; size_t reduce_tmp_floats = 0; unsigned reduce_slot = 0;   // partial sums of the two-level slab reduction
    float* fused_slab[3] = {nullptr, nullptr, nullptr}; size_t fused_slab_floats = 0;   // split-K slabs of the fused dgrad+wgrad kernels (layers 7, 6, 1)
    // use_recomp_dz: final_layer.0's dz recomputed from dlogit instead of stored (conv_fused.cuh RECOMP).  Bit-identical and
    // 268 MB less traffic each way, but measured SLOWER on MI355X (1.50 vs 1.33 ms/step): the per-element BatchNorm-backward in
    // accumulator layout costs ~20 VALU per element, and the output-conv backward is VALU-bound, not write-bound (134 us without
    // the store, 128 us with it).  Off by default; kept for the day both epilogues are cheap.
    // use_raw_wgrad: deep layers' weight gradients read MATERIALISED operands (LeakyReLU(BN(y)) / BN-backward gradient written as a
    // side effect by the kernel that stages them first) as plain copies.  Bit-identical; measured 1 % SLOWER in the step on MI355X
    // (the extra stores cost the chain more than the weight-gradient kernels gain: their time is not in the staging arithmetic).
    int use_raw_wgrad = 0;
    // use_deep: workgroup-specialised kernels of the deep layers (conv_deep.cuh).  bit 0: stride-2 conv products (dn3), bit 1: transposed
    // products (up3).  Measured on MI355X (128x128 L=16 B=256 bf16, three runs each): dn3 alone 1.262 ms/step, neither 1.267, both 1.279,
    // up3 alone 1.279 - the transposed kernel's nine LDS-DMA issues per consumer wave and K step (~130 cycles each) cost what its
    // overlap wins, so it is off by default.
    int use_deep = 1;
    // use_latent_mfma: skinny linears around the latent on the exact-f32 MFMA, one 64-feature tile x the whole batch per workgroup, no batch
    // split / slabs / reduction launches (latent_mfma.cuh).  Bits: 1 decoder_input forward, 2 its weight + bias gradient, 4 fc_mu|fc_var weight
    // (+ bias) gradient, 8 fc input gradient.  Measured on MI355X (128x128 L=16 B=256 bf16, isolated): weight gradients 22 / 26 us against
    // 30 / 28 us + 4 reductions (26 us); the forward (14 vs 12 us) and the fc input gradient (24 vs 22 us) are not faster and stay on the VALU
    // kernels - whose summation order is also the one the f32 parity gates were measured with.
    int use_latent_mfma = 6;
    int knob_skip_wgrad = 0;  // diagnostics: bit i skips the separate weight-gradient launch of BN layer i (results wrong, timing only)
    int knob_ablate_f = 0;   // diagnostics: phase ablation of conv_bwd_fused_kernel (timing only)
    // use_fused_convout: a forward with train = 2 (the fused training step) leaves the output conv, sigmoid and BCE to the
    // backward, where ONE kernel does forward and backward of that layer in one pass over y7 (conv_mfma.cuh:
    // convout_step_mfma_kernel).  convout_pending: such a forward is waiting for its backward; pending_f7: the BatchNorm
    // finalisation that kernel's prologue performs; loss_out3 / loss_kw: where vae_loss_deferred wants the ELBO scalars.
    int use_fused_convout = 1, convout_pending = 0, dlogit_valid = 0;
    // use_convout_stream: 128-pixel-wide images take the row-streaming form of that kernel (convout_stream.cuh); 0 = the tiled one
    int use_dnf_stream = 1;    // encoder.1 forward on 128x128 images: row-streaming kernel (dnfirst_stream.cuh); 0 = the tiled one
    int use_upf_stream = 1;    // row-streaming transposed-conv forward on 128x128 images (upfinal_stream.cuh): bit 0 final_layer.0, bit 1 decoder.2; 0 = the tiled kernels
    int use_fc_dgrad8 = 1;     // fc input gradient, 16-bit storage: 8 channels x 4 rows per thread with 16-byte accesses (edge_kernels.cuh: fc_dgrad8_kernel); 0 = one channel per thread
    int use_wgrad_split = 1;   // deep weight gradients: producer / consumer wave groups (wgrad_split.cuh); 0 = the 8-wave kernel
    int use_convout_stream = 1, knob_convout_bands = 0;   // (bands per image: 0 = chosen by the launcher)
    int knob_convout_step_grid = 1024;   // (= knob_convout_bwd_grid: with the same tile partition the fused kernel and convout_bwd produce bit-identical statistics)
    BnFuse pending_f7; float* loss_out3 = nullptr; float loss_kw = 0.f;
    int use_fused_wgrad = 3, knob_fused_grid = 256, use_recomp_dz = 0;   // use_fused_wgrad: bit 0 decoder (ConvT) kernels, bit 1 encoder.1 kernel
    // f16 storage: the backward runs on gradients multiplied by gmul (a power of two chosen per forward so that the stored
    // dz stay inside the f16 range: the BCE mean makes them O(1/(B*H*W))); every parameter gradient is written times ginv.
    // The backward is linear in the upstream gradient, so this changes no f32 result (powers of two are exact).  1 otherwise.
    float gmul, ginv;
    // last forward
    int B; int trained; const float* x; float *xhat, *mu, *lv, *z;
    int use_tr16, use_mfma_convout, use_pipelined, knob_up_per_cu, knob_convout_grid, knob_convout_bwd_grid, knob_down_per_cu, knob_nt_max, knob_pipe_max_cout, knob_ablate_b; long long* dbg_buf; char dbg_tag[32]; int dbg_epi; int64_t ws_bytes;
    std::vector<void*> allocs;
    // per-kernel timing (bench.py roofline): HIP events on the launch stream
    int prof; const char* tag; struct ProfRec { std::string name; hipEvent_t e0, e1; double bytes, flops; int side; int launches = 1; }; std::vector<ProfRec> prof_recs;
    hipStream_t cur_stream = nullptr; bool cur_stream_set = false;   // the caller's stream of the call in progress (profiling: tells critical-chain launches from side-stream ones)
};

// RAII: brackets the launches of one logical kernel with events when profiling is on.
struct ProfScope {
    vae_ctx* c; hipStream_t st; int idx;
    ProfScope(vae_ctx* c_, const char* name, double bytes, double flops, hipStream_t st_) : c(c_), st(st_), idx(-1) {
        if (!c || !c->prof) return;
        vae_ctx::ProfRec r; r.name = std::string(name) + (c->tag ? std::string(" @") + c->tag : std::string()); r.bytes = bytes; r.flops = flops;
        r.side = (c->cur_stream_set && st != c->cur_stream) ? 1 : 0;   // (the legacy default stream is the null handle)
        if (hipEventCreate(&r.e0) != hipSuccess || hipEventCreate(&r.e1) != hipSuccess) return;
        (void)hipEventRecord(r.e0, st);
        c->prof_recs.push_back(r); idx = (int)c->prof_recs.size() - 1;
    }
    ~ProfScope() { if (idx >= 0) (void)hipEventRecord(c->prof_recs[idx].e1, st); }
};

static inline size_t wgrad_slab_floats(const WgradKnobs& k, int B, int Hs, int Ws, int CA, int CB, int* nsplit_out, int* tps_out, int* WA_out, int* WB_out,
                                       bool wide_ok = false, bool big = false) {
    int WA, WB;
    if (wide_ok && k.wide && CA >= 128 && k.tile == 1) { WA = 4; WB = 1; }
    else if (CA >= 64 && CB >= 64 && k.tile == 0) { WA = 2; WB = 2; } else if (CA >= 64 && k.tile <= 1) { WA = 2; WB = 1; } else { WA = 1; WB = 1; }
    Tiling t = make_tiling(Hs, Ws, WG_KP);
    const int TB = 1 << t.lTB;
    const int n_tiles = ((B + TB - 1) / TB) * t.tiles_x * t.tiles_y;
    const int chan_tiles = (CA / (32 * WA)) * (CB / (32 * WB));
    const size_t per = (size_t)9 * CA * CB;
    int nsplit = std::max(1, (!big ? k.small_wgs : (WA == 4 ? k.wide_wgs : k.wgs)) / chan_tiles);
    const size_t cap = ((size_t)k.cap_mb << 20) / 4;  // bound slab traffic to 48 MiB per layer
    nsplit = (int)std::min<size_t>(nsplit, std::max<size_t>(1, cap / per));
    nsplit = std::min(nsplit, n_tiles);
    const int tps = (n_tiles + nsplit - 1) / nsplit;
    nsplit = (n_tiles + tps - 1) / tps;
    *nsplit_out = nsplit; *tps_out = tps; *WA_out = WA; *WB_out = WB;
    return per * nsplit;
}

// side streams (vae_api.hip)
struct SideFork { hipStream_t st; float* slab; int rc; };
SideFork fork_side(vae_ctx* c, hipStream_t st, int which = -1);
int join_sides(vae_ctx* c, hipStream_t st);
int join_comm(vae_ctx* c, hipStream_t st);

// entry points instantiated once per storage type (impl_bf16.hip, impl_f16.hip, impl_f32.hip)
template <typename T> int pack_weights(vae_ctx* c, const float* params, hipStream_t st);
template <typename T> int forward_impl(vae_ctx* c, const float* x, int B, const float* params, float* bn_running, int64_t* nbt,
                                       const float* eps, uint64_t seed, int train, float* xhat, float* mu, float* lv, float* z, hipStream_t st);
template <typename T> int decode_impl(vae_ctx* c, const float* z, int B, const float* params, float* bn_running, int64_t* nbt, int train,
                                      const float* x, float* xhat, hipStream_t st);
template <typename T> int backward_impl(vae_ctx* c, const float* x, const float* params, float* grads, const float* g_xhat, const float* gscale,
                                        const float* g_mu, const float* g_lv, const float* g_z, const float* g_pre, float kld_weight, int add_kl,
                                        int part, hipStream_t st);
template <typename T> int pre_latents_impl(vae_ctx* c, float* out, hipStream_t st);
template <typename T> int debug_tensor_impl(vae_ctx* c, const void* src, float* out, long n, int C, int HW, hipStream_t st);

#define VAE_INSTANTIATE(T)                                                                                                              \
    template int pack_weights<T>(vae_ctx*, const float*, hipStream_t);                                                                  \
    template int forward_impl<T>(vae_ctx*, const float*, int, const float*, float*, int64_t*, const float*, uint64_t, int, float*,     \
                                 float*, float*, float*, hipStream_t);                                                                  \
    template int decode_impl<T>(vae_ctx*, const float*, int, const float*, float*, int64_t*, int, const float*, float*, hipStream_t);  \
    template int backward_impl<T>(vae_ctx*, const float*, const float*, float*, const float*, const float*, const float*, const float*, \
                                  const float*, const float*, float, int, int, hipStream_t);                                            \
    template int pre_latents_impl<T>(vae_ctx*, float*, hipStream_t);                                                                    \
    template int debug_tensor_impl<T>(vae_ctx*, const void*, float*, long, int, int, hipStream_t);

// storage-type dispatch: VAE_DISPATCH(c->dtype, forward_impl, (c, ...))
#define VAE_DISPATCH(dtype, fn, args) ((dtype) == VAE_DTYPE_BF16 ? fn<bf16> args : (dtype) == VAE_DTYPE_F16 ? fn<f16> args : fn<float> args)
