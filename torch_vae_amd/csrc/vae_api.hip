// Context, launch sequencing and the C ABI (include/vae_step.h) of the VAE step.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>
#include <math.h>
#include <string>
#include <vector>
#include <algorithm>

#include "../../include/vae_step.h"

static thread_local std::string g_err;
static int vae_set_error(const char* what, const char* why) {
    g_err = std::string(what) + ": " + why;
    return -1;
}

#include "common.cuh"
#include "conv_mfma.cuh"
#include "conv_pipe.cuh"
#include "edge_kernels.cuh"

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

// ---------------------------------------------------------------------------
extern "C" const char* vae_last_error(void) { return g_err.c_str(); }
extern "C" int vae_abi_version(void) { return 1; }

static void param_shapes(int H, int L, int gen, int64_t* sizes) {
    const int s = gen ? H / 16 : 2;
    const int64_t F = 256LL * s * s;
    const int enc_ci[4] = {1, 32, 64, 128}, enc_co[4] = {32, 64, 128, 256};
    int k = 0;
    for (int i = 0; i < 4; ++i) { sizes[k++] = 9LL * enc_ci[i] * enc_co[i]; sizes[k++] = enc_co[i]; sizes[k++] = enc_co[i]; sizes[k++] = enc_co[i]; }
    sizes[k++] = L * F; sizes[k++] = L; sizes[k++] = L * F; sizes[k++] = L; sizes[k++] = F * L; sizes[k++] = F;
    const int dec_ci[3] = {256, 128, 64}, dec_co[3] = {128, 64, 32};
    for (int i = 0; i < 3; ++i) { sizes[k++] = 9LL * dec_ci[i] * dec_co[i]; sizes[k++] = dec_co[i]; sizes[k++] = dec_co[i]; sizes[k++] = dec_co[i]; }
    sizes[k++] = 9 * 32 * 32; sizes[k++] = 32; sizes[k++] = 32; sizes[k++] = 32; sizes[k++] = 9 * 32; sizes[k++] = 1;
}

extern "C" int vae_param_layout(int H, int L, int gen, int64_t* offsets, int64_t* sizes, int64_t* total) {
    if (H < 32 || (H & (H - 1)) || (!gen && H != 32)) return vae_set_error("vae_param_layout", "img_size must be a power of two >= 32 (exactly 32 unless generalised)");
    if (L < 4 || L % 4) return vae_set_error("vae_param_layout", "latent_dim must be a positive multiple of 4");
    param_shapes(H, L, gen, sizes);
    int64_t off = 0;
    for (int i = 0; i < VAE_NUM_PARAMS; ++i) { offsets[i] = off; off += align_up(sizes[i], 64); }
    *total = off;
    return 0;
}
extern "C" int vae_bn_layout(int64_t* offsets, int64_t* channels, int64_t* total) {
    int64_t off = 0;
    for (int i = 0; i < 8; ++i) { offsets[i] = off; channels[i] = kBnC[i]; off += 2 * kBnC[i]; }
    *total = off;
    return 0;
}

// ---------------------------------------------------------------------------
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
};

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
    hipStream_t comm; hipEvent_t ev_comm; int comm_busy;   // stream lent to the caller for the mid-backward gradient all-reduce (vae_comm_stream)
    int use_side_stream, knob_bwd_per_cu, knob_wave_nt_max, knob_lay22_min_nt, knob_conv1_grid, use_fused_bn, knob_rev, knob_lean, walk_dir, bwd_dirty, bwd_half_done;
    double* dstats; size_t n_dstats; double* accum;  // accum: [0] bce, [1] kl term, [2] sum dlogit
    // last forward
    int B; int trained; const float* x; float *xhat, *mu, *lv, *z;
    int use_tr16, use_mfma_convout, use_pipelined, knob_up_per_cu, knob_convout_grid, knob_convout_bwd_grid, knob_down_per_cu, knob_nt_max, knob_pipe_max_cout, knob_ablate_b; long long* dbg_buf; char dbg_tag[32]; int dbg_epi; int64_t ws_bytes;
    std::vector<void*> allocs;
    // per-kernel timing (bench.py roofline): HIP events on the launch stream
    int prof; const char* tag; struct ProfRec { std::string name; hipEvent_t e0, e1; double bytes, flops; }; std::vector<ProfRec> prof_recs;
};

// RAII: brackets the launches of one logical kernel with events when profiling is on.
struct ProfScope {
    vae_ctx* c; hipStream_t st; int idx;
    ProfScope(vae_ctx* c_, const char* name, double bytes, double flops, hipStream_t st_) : c(c_), st(st_), idx(-1) {
        if (!c || !c->prof) return;
        vae_ctx::ProfRec r; r.name = std::string(name) + (c->tag ? std::string(" @") + c->tag : std::string()); r.bytes = bytes; r.flops = flops;
        if (hipEventCreate(&r.e0) != hipSuccess || hipEventCreate(&r.e1) != hipSuccess) return;
        (void)hipEventRecord(r.e0, st);
        c->prof_recs.push_back(r); idx = (int)c->prof_recs.size() - 1;
    }
    ~ProfScope() { if (idx >= 0) (void)hipEventRecord(c->prof_recs[idx].e1, st); }
};

template <typename T> static T* dalloc(vae_ctx* c, size_t n) {
    void* p = nullptr;
    if (hipMalloc(&p, std::max<size_t>(n * sizeof(T), 256)) != hipSuccess) return nullptr;
    c->allocs.push_back(p); c->ws_bytes += (int64_t)std::max<size_t>(n * sizeof(T), 256);
    return reinterpret_cast<T*>(p);
}

extern "C" void vae_destroy(vae_ctx* c) {
    if (!c) return;
    for (void* p : c->allocs) (void)hipFree(p);
    if (c->n_side_ok) {
        for (int i = 0; i < vae_ctx::NSIDE; ++i) { (void)hipStreamDestroy(c->side[i]); (void)hipEventDestroy(c->ev_join[i]); }
        for (int i = 0; i < vae_ctx::NFORK; ++i) (void)hipEventDestroy(c->ev_fork[i]);
        (void)hipEventDestroy(c->ev_pack); (void)hipStreamDestroy(c->comm); (void)hipEventDestroy(c->ev_comm);
    }
    delete c;
}
extern "C" int64_t vae_workspace_bytes(const vae_ctx* c) { return c ? c->ws_bytes : 0; }

static int g_wgrad_wgs = 128, g_wgrad_cap_mb = 48, g_wgrad_tile = 1, g_wgrad_wide = 1, g_wgrad_wide_wgs = 128, g_wgrad_small_wgs = 1024, g_wgrad_mid8 = 0;
// workgroup targets: weight gradients run beside the input-gradient chain; on a saturated GPU (large batch x image) few
// workgroups keep them out of its way (-4 % step time at the bench workload), a small problem wants them everywhere.
//   // wide: 128x32-channel tiles on 8 waves where the low-res side has >= 128 channels (bf16 prefetching kernel)
//   // tile 1: 64x32 channel tiles (prefetching kernel) also where 64x64 would fit   // split-K sizing (vae_set_option knobs; slabs are sized at vae_create for the defaults)
static size_t wgrad_slab_floats(int B, int Hs, int Ws, int CA, int CB, int* nsplit_out, int* tps_out, int* WA_out, int* WB_out, bool wide_ok = false, bool big = false) {
    int WA, WB;
    if (wide_ok && g_wgrad_wide && CA >= 128 && g_wgrad_tile == 1) { WA = 4; WB = 1; }
    else if (CA >= 64 && CB >= 64 && g_wgrad_tile == 0) { WA = 2; WB = 2; } else if (CA >= 64 && g_wgrad_tile <= 1) { WA = 2; WB = 1; } else { WA = 1; WB = 1; }
    Tiling t = make_tiling(Hs, Ws, WG_KP);
    const int TB = 1 << t.lTB;
    const int n_tiles = ((B + TB - 1) / TB) * t.tiles_x * t.tiles_y;
    const int chan_tiles = (CA / (32 * WA)) * (CB / (32 * WB));
    const size_t per = (size_t)9 * CA * CB;
    int nsplit = std::max(1, (!big ? g_wgrad_small_wgs : (WA == 4 ? g_wgrad_wide_wgs : g_wgrad_wgs)) / chan_tiles);
    const size_t cap = ((size_t)g_wgrad_cap_mb << 20) / 4;  // bound slab traffic to 48 MiB per layer
    nsplit = (int)std::min<size_t>(nsplit, std::max<size_t>(1, cap / per));
    nsplit = std::min(nsplit, n_tiles);
    const int tps = (n_tiles + nsplit - 1) / nsplit;
    nsplit = (n_tiles + tps - 1) / tps;
    *nsplit_out = nsplit; *tps_out = tps; *WA_out = WA; *WB_out = WB;
    return per * nsplit;
}

extern "C" vae_ctx* vae_create(int H, int L, int maxB, int dtype, int gen) {
    vae_ctx* c = new vae_ctx();
    c->H = H; c->L = L; c->maxB = maxB; c->dtype = dtype; c->gen = gen; c->ws_bytes = 0; c->use_tr16 = 1; c->use_mfma_convout = 1; c->use_pipelined = 1; c->knob_up_per_cu = 4; c->knob_convout_grid = 1536; c->knob_convout_bwd_grid = 1536; c->knob_down_per_cu = 2; c->knob_nt_max = 4; c->knob_pipe_max_cout = 256; c->knob_ablate_b = 0; c->use_side_stream = 1; c->knob_bwd_per_cu = 0; c->knob_wave_nt_max = 4; c->knob_lay22_min_nt = 4; c->knob_conv1_grid = 1024; c->use_fused_bn = 1; c->knob_rev = 4; c->knob_lean = 1; c->walk_dir = 0; c->n_side_ok = 0; c->side_rr = 0; c->fork_rr = 0; c->comm_busy = 0; c->dbg_buf = nullptr; c->dbg_tag[0] = 0; c->dbg_epi = 0;
    if (getenv("VAE_NO_SIDE_STREAM")) c->use_side_stream = 0;   // diagnostics: everything on the caller's stream
    c->packed_for = nullptr; c->bwd_dirty = 1; c->bwd_half_done = 0; c->B = 0; c->trained = 0; c->prof = 0; c->tag = nullptr;
    if (vae_param_layout(H, L, gen, c->poff, c->psz, &c->ptotal) != 0) { delete c; return nullptr; }
    if (dtype != VAE_DTYPE_F32 && dtype != VAE_DTYPE_BF16) { vae_set_error("vae_create", "bad dtype"); delete c; return nullptr; }
    if (maxB < 1) { vae_set_error("vae_create", "max_batch < 1"); delete c; return nullptr; }
    vae_bn_layout(c->bnoff, c->bnc, &c->bntotal);
    c->s = gen ? H / 16 : 2; c->s2 = c->s * c->s; c->F = 256LL * c->s2;
    c->npad_fc = (int)align_up(2 * L, 32); c->npad_di = (int)align_up(L, 32);
    c->esz = dtype == VAE_DTYPE_BF16 ? 2 : 4;
    const size_t B = maxB;
    // BN'd tensors: encoder outputs H/2..H/16, decoder outputs 2s..8s, final convT output H.
    const int hs[8] = {H / 2, H / 4, H / 8, H / 16, 2 * c->s, 4 * c->s, 8 * c->s, 16 * c->s};
    size_t nd = 0;
    for (int i = 0; i < 8; ++i) nd += 4 * kBnC[i];
    nd *= STAT_R;                       // replicas (common.cuh: STAT_R)
    c->n_dstats = nd + 8 * STAT_R;
    c->dstats = dalloc<double>(c, c->n_dstats);
    bool ok = c->dstats != nullptr;
    double* dp = c->dstats;
    for (int i = 0; i < 8 && ok; ++i) {
        BnLayer& l = c->lay[i];
        l.C = kBnC[i]; l.H = hs[i]; l.W = hs[i];
        l.stat_f = dp; dp += 2 * l.C * STAT_R;
        const size_t n = B * l.H * l.W * l.C;
        l.y = dalloc<char>(c, n * c->esz); l.dz = dalloc<char>(c, n * c->esz); l.block = dalloc<float>(c, LC_ROWS * l.C);
        ok = l.y && l.dz && l.block;
        const int base = i < 4 ? 4 * i : (i < 7 ? 22 + 4 * (i - 4) : 34);
        l.p_convw = base; l.p_convb = base + 1; l.p_gamma = base + 2; l.p_beta = base + 3;
    }
    for (int i = 0; i < 8; ++i) { c->lay[i].stat_b = dp; dp += 2 * kBnC[i] * STAT_R; }
    c->accum = dp;
    if (ok) {
        c->d0 = dalloc<char>(c, B * c->F * c->esz); c->dd0 = dalloc<char>(c, B * c->F * c->esz);
        c->eps = dalloc<float>(c, B * L); c->dlat = dalloc<float>(c, B * 2 * L);
        c->dlogit = dalloc<float>(c, B * H * H); c->dlogit2 = dalloc<float>(c, B * H * H); c->ident = dalloc<float>(c, 3 * 256); c->wout_t = dalloc<float>(c, 288);
        ok = c->d0 && c->dd0 && c->eps && c->dlat && c->dlogit && c->dlogit2 && c->ident && c->wout_t;
    }
    // packed weight images
    const int ci[8] = {1, 32, 64, 128, 256, 128, 64, 32}, co[8] = {32, 64, 128, 256, 128, 64, 32, 32};
    for (int i = 1; i < 8 && ok; ++i) {
        const size_t n = (size_t)9 * ci[i] * co[i];
        c->wp_fwd[i] = dalloc<char>(c, n * c->esz); c->wp_dg[i] = dalloc<char>(c, n * c->esz);
        ok = c->wp_fwd[i] && c->wp_dg[i];
    }
    if (ok) {
        c->fcpack = dalloc<char>(c, (size_t)c->F * c->npad_fc * c->esz);
        c->dipack = dalloc<char>(c, (size_t)c->F * c->npad_di * c->esz);
        c->d_descs = dalloc<PackDesc>(c, 32);
        ok = c->fcpack && c->dipack && c->d_descs;
    }
    // slab: max over all split-K users
    size_t slab = 2048 * 288;  // conv1 wgrad / convout bwd: up to 2048 workgroups x 288
    if (ok) {
        int a, b2, wa, wb;
        for (int i = 1; i < 4; ++i) slab = std::max(slab, wgrad_slab_floats(maxB, c->lay[i].H, c->lay[i].W, co[i], ci[i], &a, &b2, &wa, &wb, dtype == VAE_DTYPE_BF16));
        for (int i = 4; i < 8; ++i) slab = std::max(slab, wgrad_slab_floats(maxB, c->lay[i].H / 2, c->lay[i].W / 2, ci[i], co[i], &a, &b2, &wa, &wb, dtype == VAE_DTYPE_BF16));
        const size_t ksteps = c->F / 16;
        slab = std::max(slab, (size_t)std::min<size_t>(ksteps, 512) * maxB * c->npad_fc);
        slab = std::max(slab, (size_t)std::min<size_t>(ksteps, 512) * maxB * c->npad_di);
        slab = std::max(slab, (size_t)8 * (2 * (size_t)L * c->F + c->F));   // fc / decoder_input weight-gradient batch slices
        c->slab_floats = slab;
        c->slab = dalloc<float>(c, slab);
        ok = c->slab != nullptr;
        for (int i = 0; i < vae_ctx::NSIDE && ok; ++i) { c->side_slab[i] = dalloc<float>(c, slab); ok = c->side_slab[i] != nullptr; }
        if (ok) {
            // side-stream priority: VAE_SIDE_PRIORITY=low|high (default: the device's default priority)
            int prio_least = 0, prio_greatest = 0, side_prio = 0;
            (void)hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest);
            if (const char* e = getenv("VAE_SIDE_PRIORITY")) side_prio = !strcmp(e, "low") ? prio_least : !strcmp(e, "high") ? prio_greatest : 0;
            for (int i = 0; i < vae_ctx::NSIDE && ok; ++i)
                ok = hipStreamCreateWithPriority(&c->side[i], hipStreamNonBlocking, side_prio) == hipSuccess && hipEventCreateWithFlags(&c->ev_join[i], hipEventDisableTiming) == hipSuccess;
            for (int i = 0; i < vae_ctx::NFORK && ok; ++i) ok = hipEventCreateWithFlags(&c->ev_fork[i], hipEventDisableTiming) == hipSuccess;
            if (ok) ok = hipEventCreateWithFlags(&c->ev_pack, hipEventDisableTiming) == hipSuccess &&
                         hipStreamCreateWithFlags(&c->comm, hipStreamNonBlocking) == hipSuccess &&
                         hipEventCreateWithFlags(&c->ev_comm, hipEventDisableTiming) == hipSuccess;
            c->n_side_ok = ok ? 1 : 0;
        }
    }
    if (!ok) { vae_set_error("vae_create", "hipMalloc failed"); vae_destroy(c); return nullptr; }
    std::vector<float> id(3 * 256, 0.f);
    for (int i = 0; i < 256; ++i) id[i] = 1.f;
    if (hipMemcpy(c->ident, id.data(), id.size() * 4, hipMemcpyHostToDevice) != hipSuccess) { vae_set_error("vae_create", "memcpy failed"); vae_destroy(c); return nullptr; }
    return c;
}

extern "C" int vae_set_option(vae_ctx* c, const char* name, int value) {
    if (!c) return vae_set_error("vae_set_option", "null ctx");
    if (!strcmp(name, "use_tr16")) { c->use_tr16 = value; return 0; }
    if (!strcmp(name, "use_mfma_convout")) { c->use_mfma_convout = value; return 0; }
    if (!strcmp(name, "use_pipelined")) { c->use_pipelined = value; return 0; }
    if (!strcmp(name, "knob_up_per_cu")) { c->knob_up_per_cu = value; return 0; }
    if (!strcmp(name, "knob_convout_grid")) { c->knob_convout_grid = value; return 0; }
    if (!strcmp(name, "knob_down_per_cu")) { c->knob_down_per_cu = std::max(1, value); return 0; }
    if (!strcmp(name, "knob_convout_bwd_grid")) { c->knob_convout_bwd_grid = value; return 0; }
    if (!strcmp(name, "knob_nt_max")) { c->knob_nt_max = value; return 0; }
    if (!strcmp(name, "knob_pipe_max_cout")) { c->knob_pipe_max_cout = value; return 0; }
    if (!strcmp(name, "knob_ablate_b")) { c->knob_ablate_b = value; return 0; }
    if (!strcmp(name, "use_side_stream")) { c->use_side_stream = value; return 0; }
    if (!strcmp(name, "knob_bwd_per_cu")) { c->knob_bwd_per_cu = value; return 0; }
    if (!strcmp(name, "knob_wave_nt_max")) { c->knob_wave_nt_max = value; return 0; }
    if (!strcmp(name, "use_fused_bn")) { c->use_fused_bn = value; return 0; }
    if (!strcmp(name, "knob_lay22_min_nt")) { c->knob_lay22_min_nt = value; return 0; }
    if (!strcmp(name, "knob_conv1_grid")) { c->knob_conv1_grid = value; return 0; }
    if (!strcmp(name, "knob_rev")) { c->knob_rev = value; return 0; }
    if (!strcmp(name, "knob_lean")) { c->knob_lean = value; return 0; }   // bit 0: noise beside conv1; 1: BN backward inside conv1_wgrad; 2: deferred loss on a side stream
    if (!strcmp(name, "knob_wgrad_tile")) { g_wgrad_tile = value; return 0; }
    if (!strcmp(name, "knob_wgrad_wide")) { g_wgrad_wide = value; return 0; }
    if (!strcmp(name, "knob_wgrad_mid8")) { g_wgrad_mid8 = value; return 0; }
    if (!strcmp(name, "knob_wgrad_wide_wgs")) { g_wgrad_wide_wgs = std::min(value, 1024); return 0; }
    if (!strcmp(name, "knob_wgrad_wgs")) { g_wgrad_wgs = std::min(value, 1024); return 0; }
    if (!strcmp(name, "knob_wgrad_cap_mb")) { g_wgrad_cap_mb = std::min(value, 48); return 0; }
    return vae_set_error("vae_set_option", "unknown option");
}

// ---------------------------------------------------------------------------
template <typename K> static int set_lds(K kernel, size_t bytes) {
    if (bytes > 160 * 1024) return vae_set_error("lds", "tile needs more than 160 KiB LDS");
    if (bytes > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
        if (e != hipSuccess) return vae_set_error("hipFuncSetAttribute", hipGetErrorString(e));
    }
    return 0;
}

template <typename T> static int launch_conv_pipe(vae_ctx* c, ConvArgs<T> a, bool is_down, hipStream_t st);
// the pipelined kernels index their tensors with 32-bit byte offsets (and signed 32-bit element offsets)
template <typename T> static bool fits_i32(const ConvArgs<T>& a) { return 4.0 * a.B * a.Hs * a.Ws * std::max(a.Cin, a.Cout) * sizeof(T) < 4294967296.0 && 4.0 * a.B * a.Hs * a.Ws * std::max(a.Cin, a.Cout) < 2147483648.0; }

template <typename T>
static int launch_down(vae_ctx* c, ConvArgs<T> a, hipStream_t st) {
    if (c->use_pipelined && a.Cout <= c->knob_pipe_max_cout && fits_i32(a)) return launch_conv_pipe<T>(c, a, true, st);
    Tiling t = make_tiling(a.Hs, a.Ws, 128);
    a.lth = t.lth; a.ltw = t.ltw; a.lTB = t.lTB; a.tiles_x = t.tiles_x; a.tiles_y = t.tiles_y;
    const int TB = 1 << t.lTB, th = 1 << t.lth, tw = 1 << t.ltw;
    const int n_tiles = ((a.B + TB - 1) / TB) * t.tiles_x * t.tiles_y;
    a.m_pp = fastdiv_magic((2 * th + 1) * (2 * tw + 1)); a.m_pw = fastdiv_magic(2 * tw + 1); a.m_tx = fastdiv_magic(t.tiles_x); a.m_txy = fastdiv_magic(t.tiles_x * t.tiles_y);
    const int NT = std::min(4, a.Cout / 32);
    const size_t lds = ((3 * a.Cin * 4 + 15) & ~15) + (size_t)TB * (2 * th + 1) * (2 * tw + 1) * PATCH_PITCH + 4 * NT * 32 * 2 * 4;
    dim3 grid(n_tiles, a.Cout / (32 * NT));
    const double px_out = (double)a.B * a.Hs * a.Ws, px_in = 4 * px_out;
    ProfScope ps(c, a.epi == EPI_FWD ? "down_fwd(conv)" : "down_bwd(convT dgrad)",
                 sizeof(T) * (px_in * a.Cin * (a.two_src ? 2 : 1) + px_out * a.Cout * (a.epi == EPI_BWD ? 2 : 1) + 9.0 * a.Cin * a.Cout),
                 2.0 * 9 * a.Cin * a.Cout * px_out, st);
#define DOWN_CASE(N) { if (set_lds(down_kernel<T, N>, lds)) return -1; hipLaunchKernelGGL((down_kernel<T, N>), grid, dim3(256), lds, st, a); }
    if (NT == 1) DOWN_CASE(1) else if (NT == 2) DOWN_CASE(2) else DOWN_CASE(4)
#undef DOWN_CASE
    LAUNCH_CHECK("down_kernel");
    return 0;
}

template <typename T>
static int launch_up(vae_ctx* c, ConvArgs<T> a, hipStream_t st) {
    if (c->use_pipelined && a.Cout <= c->knob_pipe_max_cout && fits_i32(a)) return launch_conv_pipe<T>(c, a, false, st);
    Tiling t = make_tiling(a.Hs, a.Ws, 128);
    a.lth = t.lth; a.ltw = t.ltw; a.lTB = t.lTB; a.tiles_x = t.tiles_x; a.tiles_y = t.tiles_y;
    const int TB = 1 << t.lTB, th = 1 << t.lth, tw = 1 << t.ltw;
    const int n_tiles = ((a.B + TB - 1) / TB) * t.tiles_x * t.tiles_y;
    a.m_pp = fastdiv_magic((th + 1) * (tw + 1)); a.m_pw = fastdiv_magic(tw + 1); a.m_tx = fastdiv_magic(t.tiles_x); a.m_txy = fastdiv_magic(t.tiles_x * t.tiles_y);
    const int NT = std::min(2, a.Cout / 32);
    const size_t lds = ((3 * a.Cin * 4 + 15) & ~15) + (size_t)TB * (th + 1) * (tw + 1) * PATCH_PITCH + 4 * NT * 32 * 2 * 4;
    dim3 grid(n_tiles, a.Cout / (32 * NT));
    const double px_in = (double)a.B * a.Hs * a.Ws, px_out = 4 * px_in;
    ProfScope ps(c, a.epi == EPI_FWD ? "up_fwd(convT)" : "up_bwd(conv dgrad)",
                 sizeof(T) * (px_in * a.Cin * (a.two_src ? 2 : 1) + px_out * a.Cout * (a.epi == EPI_BWD ? 2 : 1) + 9.0 * a.Cin * a.Cout),
                 2.0 * 9 * a.Cin * a.Cout * px_in, st);
#define UP_CASE(N) { if (set_lds(up_kernel<T, N>, lds)) return -1; hipLaunchKernelGGL((up_kernel<T, N>), grid, dim3(256), lds, st, a); }
    if (NT == 1) UP_CASE(1) else UP_CASE(2)
#undef UP_CASE
    LAUNCH_CHECK("up_kernel");
    return 0;
}

// persistent, prefetched variants (conv_pipe.cuh)
template <typename T>
static int launch_conv_pipe(vae_ctx* c, ConvArgs<T> a, bool is_down, hipStream_t st) {
    // register budget: two-source (gradient) loads and the 4-parity accumulators of `up` keep NT at 1
    int NT = std::min(c->knob_nt_max, a.Cout / 32);
    NT = NT >= 4 ? 4 : (NT >= 2 ? 2 : 1);
    if (!is_down || sizeof(T) == 4) NT = std::min(NT, 2);
    if (!is_down && (a.epi == EPI_BWD || sizeof(T) == 4)) NT = 1;
    // tile organisation: 2x2 wave grid over a 128-pixel workgroup tile (wide down tiles: halves the weight-fragment
    // traffic), wave-independent 32-pixel tiles (no workgroup barrier in the loop), or one row of waves per workgroup tile
    const bool lay22 = is_down && NT >= 2 && NT >= c->knob_lay22_min_nt;   // (f32: NT is 2, used by the exact-arithmetic tests)
    const bool wv = sizeof(T) == 2 && !lay22 && NT <= c->knob_wave_nt_max;
    Tiling t = make_tiling(a.Hs, a.Ws, wv ? 32 : 128);
    a.lth = t.lth; a.ltw = t.ltw; a.lTB = t.lTB; a.tiles_x = t.tiles_x; a.tiles_y = t.tiles_y;
    const int TB = 1 << t.lTB, th = 1 << t.lth, tw = 1 << t.ltw;
    const int n_mt = ((a.B + TB - 1) / TB) * t.tiles_x * t.tiles_y;
    const int PHW = is_down ? (2 * th + 1) * (2 * tw + 1) : (th + 1) * (tw + 1);
    a.m_pp = fastdiv_magic(PHW); a.m_pw = fastdiv_magic(is_down ? 2 * tw + 1 : tw + 1);
    a.m_tx = fastdiv_magic(t.tiles_x); a.m_txy = fastdiv_magic(t.tiles_x * t.tiles_y);
    const int ntn = a.Cout / (32 * NT), n_pairs = n_mt * ntn;
    a.n_mt = n_mt; a.rev = ((c->knob_rev >> 2) & 1) ? ((a.epi == EPI_FWD) ? ((c->knob_rev >> 4) & 1) : 1) : 0;   // bit 2: backward launches, bit 4: forward too
    if (c->knob_rev & 32) { a.rev = c->walk_dir; c->walk_dir ^= 1; }   // bit 5: alternate the direction launch by launch
    const size_t opitch = 32 * NT * sizeof(T) + 16;
    const size_t lds = ((3 * a.Cin * 4 + 15) & ~15) + (size_t)(wv ? 4 : 1) * TB * PHW * PATCH_PITCH + (lay22 ? 256 * (16 * NT * sizeof(T) + 16) : (is_down ? 128 : 256) * opitch) + 4 * NT * 32 * 2 * 4 +
                       std::max<size_t>((size_t)TB * PHW * 4, (size_t)(is_down ? 10 : 3) * (wv ? 64 : 256)) * 8;   // + the per-item staging table (padded to MAXI*SSTR)
    if (lds > 160 * 1024) return vae_set_error("conv_pipe", "tile does not fit LDS");
    if (c->knob_ablate_b) a.two_src |= 2;
    a.dbg = (c->dbg_buf && is_down == !(c->dbg_epi & 16) && c->tag && !strcmp(c->tag, c->dbg_tag) && a.epi == (c->dbg_epi & 15)) ? c->dbg_buf : nullptr;
    if ((a.two_src & 1) && a.slope != 1.f) return vae_set_error("conv_pipe", "gradient operands are loaded without LeakyReLU (slope must be 1)");
    const int per_cu = (int)std::max<size_t>(1, std::min<size_t>(is_down ? c->knob_down_per_cu : c->knob_up_per_cu, (160 * 1024) / lds));
    const int n_wg_pairs = wv ? ((n_mt + 3) / 4) * ntn : n_pairs;    // workgroup-level work items
    int grid = std::min(n_wg_pairs, 256 * ((c->knob_bwd_per_cu > 0 && a.epi != EPI_FWD) ? std::min(per_cu, c->knob_bwd_per_cu) : per_cu));
    grid = std::max(ntn, grid / ntn * ntn);   // a workgroup must stay on one N tile (register-resident statistics)
    const double px_lo = (double)a.B * a.Hs * a.Ws, px_hi = 4 * px_lo;
    const double px_in = is_down ? px_hi : px_lo, px_out = is_down ? px_lo : px_hi;
    ProfScope ps(c, is_down ? (a.epi == EPI_FWD ? "down_fwd(conv)" : "down_bwd(convT dgrad)") : (a.epi == EPI_FWD ? "up_fwd(convT)" : "up_bwd(conv dgrad)"),
                 sizeof(T) * (px_in * a.Cin * ((a.two_src & 1) ? 2 : 1) + px_out * a.Cout * (a.epi == EPI_BWD ? 2 : 1) + 9.0 * a.Cin * a.Cout),
                 2.0 * 9 * a.Cin * a.Cout * px_lo, st);
    if (((a.two_src & 1) != 0) != (a.epi != EPI_FWD)) return vae_set_error("conv_pipe", "forward launches stage one source, backward launches two");
#define PIPE_CASE(K, N, E, V) { if (set_lds(K<T, N, E, V>, lds)) return -1; hipLaunchKernelGGL((K<T, N, E, V>), dim3(grid), dim3(256), lds, st, a, n_pairs, ntn); }
#define PIPE_CASE22(N, E) { if (set_lds(down2_kernel<T, N, E, false, 1>, lds)) return -1; hipLaunchKernelGGL((down2_kernel<T, N, E, false, 1>), dim3(grid), dim3(256), lds, st, a, n_pairs, ntn); }
#define PIPE_WV(K, N, E) { if constexpr (sizeof(T) == 2) { if (wv) PIPE_CASE(K, N, E, true) else PIPE_CASE(K, N, E, false) } else PIPE_CASE(K, N, E, false) }
#define PIPE_EPI(K, N) { if (a.epi == EPI_FWD) PIPE_WV(K, N, EPI_FWD) else if (a.epi == EPI_BWD) PIPE_WV(K, N, EPI_BWD) else PIPE_WV(K, N, EPI_PLAIN) }
#define PIPE_EPI22(N) { if (a.epi == EPI_FWD) PIPE_CASE22(N, EPI_FWD) else if (a.epi == EPI_BWD) PIPE_CASE22(N, EPI_BWD) else PIPE_CASE22(N, EPI_PLAIN) }
    if (lay22) { if (NT == 2) PIPE_EPI22(2) else { if constexpr (sizeof(T) == 2) PIPE_EPI22(4) } }
    else if (is_down) { if (NT == 1) PIPE_EPI(down2_kernel, 1) else if (NT == 2) PIPE_EPI(down2_kernel, 2) else PIPE_EPI(down2_kernel, 4) }
    else if (a.epi == EPI_FWD) { if (NT == 1) PIPE_WV(up2_kernel, 1, EPI_FWD) else PIPE_WV(up2_kernel, 2, EPI_FWD) }
    else if (a.epi == EPI_BWD) PIPE_WV(up2_kernel, 1, EPI_BWD)
    else return vae_set_error("conv_pipe", "up kernel has no plain epilogue");
#undef PIPE_EPI22
#undef PIPE_EPI
#undef PIPE_WV
#undef PIPE_CASE22
#undef PIPE_CASE
    LAUNCH_CHECK("conv_pipe_kernel");
    return 0;
}

static int launch_reduce(const float* slab, int nslab, size_t n, float* out, int CA, int CB, hipStream_t st, vae_ctx* c = nullptr) {
    ProfScope ps(c, "reduce_slab", 4.0 * n * (nslab + 1), 0, st);
    hipLaunchKernelGGL(reduce_slab_kernel, dim3((unsigned)((n + 63) / 64)), dim3(256), 0, st, slab, nslab, (int)n, out, CA, CB);
    LAUNCH_CHECK("reduce_slab_kernel");
    return 0;
}

template <typename T>
static int launch_wgrad(vae_ctx* c, WgradArgs<T> a, float* dw_out, hipStream_t st, float* slab_buf = nullptr) {
    if (!slab_buf) slab_buf = c->slab;
    int nsplit, tps, WA, WB;
    const bool big = (double)c->B * c->H * c->H >= (double)(1 << 21);   // e.g. 128x128 at batch >= 128
    const size_t need = wgrad_slab_floats(a.B, a.Hs, a.Ws, a.CA, a.CB, &nsplit, &tps, &WA, &WB, c->use_pipelined && sizeof(T) == 2, big);
    if (need > c->slab_floats) return vae_set_error("wgrad", "slab too small");
    Tiling t = make_tiling(a.Hs, a.Ws, WG_KP);
    a.lth = t.lth; a.ltw = t.ltw; a.lTB = t.lTB; a.tiles_x = t.tiles_x; a.tiles_y = t.tiles_y;
    const int TB = 1 << t.lTB, th = 1 << t.lth, tw = 1 << t.ltw;
    a.n_tiles = ((a.B + TB - 1) / TB) * t.tiles_x * t.tiles_y; a.tiles_per_split = tps;
    a.slab = slab_buf; a.use_tr16 = c->use_tr16; a.rev = (c->knob_rev >> 3) & 1;
    a.m_pp = fastdiv_magic((2 * th + 1) * (2 * tw + 1)); a.m_pw = fastdiv_magic(2 * tw + 1); a.m_tx = fastdiv_magic(t.tiles_x); a.m_txy = fastdiv_magic(t.tiles_x * t.tiles_y);
    const bool mid8 = g_wgrad_mid8 && WA == 2 && WB == 1 && c->use_pipelined && sizeof(T) == 2;   // eight waves on the 64x32-channel tile
    const int nthr = (WA == 4 || mid8) ? 512 : 256, maxg = (5 * WB * 256 + nthr - 1) / nthr;
    const size_t lds = (size_t)(3 * 32 * WA + 3 * 32 * WB) * 4 + (size_t)WG_KP * (32 * WA * sizeof(T) + 16) +
                       (size_t)TB * (2 * th + 1) * (2 * tw + 1) * (32 * WB * sizeof(T) + 16) +
                       ((c->use_pipelined && sizeof(T) == 2) ? std::max<size_t>((size_t)TB * (2 * th + 1) * (2 * tw + 1) * (32 * WB * sizeof(T) / 16), (size_t)maxg * nthr) * 8 : 0);   // + staging table (prefetching variants, padded to MAXG*threads)
    dim3 grid(nsplit, a.CA / (32 * WA), a.CB / (32 * WB));
    const double px_s = (double)a.B * a.Hs * a.Ws;
    {
    ProfScope ps(c, "wgrad_kernel",
                 sizeof(T) * (px_s * a.CA * (a.s_two ? 2 : 1) + 4 * px_s * a.CB * (a.g_two ? 2 : 1)) + 4.0 * 9 * a.CA * a.CB,
                 2.0 * 9 * a.CA * a.CB * px_s, st);
    // s_two/g_two identify the layer kind: Conv2d (gradient on the low-res side) or ConvTranspose2d
    if (a.s_two == a.g_two) return vae_set_error("wgrad", "exactly one operand must be the gradient");
    const bool convt = a.g_two != 0, pre = c->use_pipelined && sizeof(T) == 2;
#define WG_CASE(A_, B_, C_, P_) { if (set_lds(wgrad_kernel<T, A_, B_, C_, P_>, lds)) return -1; hipLaunchKernelGGL((wgrad_kernel<T, A_, B_, C_, P_>), grid, dim3(256), lds, st, a); }
#define WG_KIND(A_, B_, P_) { if (convt) WG_CASE(A_, B_, true, P_) else WG_CASE(A_, B_, false, P_) }
    if (WA == 4) {
        if constexpr (sizeof(T) == 2) {
            if (convt) { if (set_lds(wgrad_kernel<T, 4, 1, true, true, 8>, lds)) return -1; hipLaunchKernelGGL((wgrad_kernel<T, 4, 1, true, true, 8>), grid, dim3(512), lds, st, a); }
            else { if (set_lds(wgrad_kernel<T, 4, 1, false, true, 8>, lds)) return -1; hipLaunchKernelGGL((wgrad_kernel<T, 4, 1, false, true, 8>), grid, dim3(512), lds, st, a); }
        }
    }
    else if (mid8) {
        if constexpr (sizeof(T) == 2) {
            if (convt) { if (set_lds(wgrad_kernel<T, 2, 1, true, true, 8>, lds)) return -1; hipLaunchKernelGGL((wgrad_kernel<T, 2, 1, true, true, 8>), grid, dim3(512), lds, st, a); }
            else { if (set_lds(wgrad_kernel<T, 2, 1, false, true, 8>, lds)) return -1; hipLaunchKernelGGL((wgrad_kernel<T, 2, 1, false, true, 8>), grid, dim3(512), lds, st, a); }
        }
    }
    else if (WA == 2 && WB == 2) { if (pre) WG_KIND(2, 2, true) else WG_KIND(2, 2, false) }
    else if (WA == 2 && WB == 1) { if (pre) WG_KIND(2, 1, true) else WG_KIND(2, 1, false) }
    else { if (pre) WG_KIND(1, 1, true) else WG_KIND(1, 1, false) }
#undef WG_KIND
#undef WG_CASE
    LAUNCH_CHECK("wgrad_kernel");
    }
    return launch_reduce(slab_buf, nsplit, (size_t)9 * a.CA * a.CB, dw_out, a.CA, a.CB, st, c);
}

template <typename T>
static int launch_dense(vae_ctx* c, DenseArgs<T> a, int* nsplit_out, hipStream_t st) {
    const int NT = std::min(4, a.Npad / 32);
    const int mt = (a.M + 127) / 128, ntile = a.Npad / (32 * NT), ksteps = a.K / 16;
    int nsplit = std::max(1, std::min(ksteps, 512 / std::max(1, mt * ntile)));
    a.ksteps_per_split = (ksteps + nsplit - 1) / nsplit;
    nsplit = (ksteps + a.ksteps_per_split - 1) / a.ksteps_per_split;
    if ((size_t)nsplit * a.M * a.Npad > c->slab_floats) return vae_set_error("dense", "slab too small");
    a.slab = c->slab;
    dim3 grid(mt, nsplit, ntile);
    ProfScope ps(c, "dense(fc / decoder_input dgrad)", sizeof(T) * ((double)a.M * a.K + (double)a.K * a.Npad), 2.0 * a.M * a.K * a.Npad, st);
    if (NT == 1) hipLaunchKernelGGL((dense_kernel<T, 1>), grid, dim3(256), 0, st, a);
    else if (NT == 2) hipLaunchKernelGGL((dense_kernel<T, 2>), grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL((dense_kernel<T, 4>), grid, dim3(256), 0, st, a);
    LAUNCH_CHECK("dense_kernel");
    *nsplit_out = nsplit;
    return 0;
}

// ---------------------------------------------------------------------------
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
__global__ void counter_normal_kernel(float* out, long n, unsigned long long seed, unsigned long long stream) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double u1 = counter_uniform(i, seed, 2 * stream + 1000003ULL), u2 = counter_uniform(i, seed, 2 * stream + 1000004ULL);
    out[i] = (float)(sqrt(-2.0 * log(1.0 - u1)) * cos(2.0 * 3.14159265358979323846 * u2));
}
// data_generators.py:45-77 restated with the counter generator (stream 777); one workgroup per image
__global__ void synth_pianoroll_kernel(float* x, int H, unsigned long long seed, int max_lines) {
    const int b = blockIdx.x;
    const int width = 1 + (int)(counter_uniform(0, seed, 777) * 4);
    const unsigned long long base = 1 + (unsigned long long)b * (1 + 4 * max_lines);
    const int n_lines = 1 + (int)(counter_uniform(base, seed, 777) * max_lines);
    for (int p = threadIdx.x; p < H * H; p += blockDim.x) {
        const int py = p / H, px = p % H;
        float v = 0.f;
        for (int li = 0; li < n_lines; ++li) {
            const unsigned long long k = base + 1 + 4ULL * li;
            const bool vert = counter_uniform(k, seed, 777) < 0.5;
            const int pos = (int)(counter_uniform(k + 1, seed, 777) * H);
            const int start = (int)(counter_uniform(k + 2, seed, 777) * H);
            const int end = start + (int)(counter_uniform(k + 3, seed, 777) * (H - start));
            const int lo = max(0, pos - width / 2), hi = min(H, pos + width / 2 + 1);
            const int along = vert ? py : px, across = vert ? px : py;
            if (along >= start && along < end && across >= lo && across < hi) v = 1.f;
        }
        x[(size_t)b * H * H + p] = v;
    }
}
extern "C" int vae_synth_pianoroll(float* x, int B, int H, uint64_t seed, vae_stream_t stream) {
    hipLaunchKernelGGL(synth_pianoroll_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, x, H, (unsigned long long)seed, 20);
    LAUNCH_CHECK("synth_pianoroll_kernel");
    return 0;
}

// o[0] = sum over the STAT_R replicas of an accumulator slot
__global__ void accum_to_f32_kernel(const double* slot, float* o) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        double s = 0.0;
        for (int rep = 0; rep < STAT_R; ++rep) s += slot[rep * 8];
        o[0] = (float)s;
    }
}
__global__ void d2f_kernel(const double* s, float* o, int n, float scale) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) o[i] = (float)(s[i] * scale);
}
__global__ void zero_f32_kernel(float* o, long n) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) o[i] = 0.f;
}

// ---------------------------------------------------------------------------
template <typename T>
static int pack_weights(vae_ctx* c, const float* params, hipStream_t st) {
    std::vector<PackDesc>& d = c->h_descs;
    if (c->packed_for != params || d.empty()) {
        d.clear();
        const int ci[8] = {1, 32, 64, 128, 256, 128, 64, 32}, co[8] = {32, 64, 128, 256, 128, 64, 32, 32};
        for (int i = 1; i < 8; ++i) {
            const bool conv = i < 4;  // Conv2d [co][ci][9] vs ConvTranspose2d [ci][co][9]
            PackDesc p; memset(&p, 0, sizeof(p));
            p.src = params + c->poff[c->lay[i].p_convw]; p.kind = 0; p.n = 9L * ci[i] * co[i];
            p.A = conv ? co[i] : ci[i]; p.Bc = conv ? ci[i] : co[i];
            p.dst = c->wp_fwd[i]; p.k_is_first = conv ? 0 : 1; d.push_back(p);   // K = ci
            p.dst = c->wp_dg[i]; p.k_is_first = conv ? 1 : 0; d.push_back(p);    // K = co
        }
        PackDesc p; memset(&p, 0, sizeof(p));
        p.kind = 1; p.src = params + c->poff[16]; p.src2 = params + c->poff[18]; p.dst = c->fcpack; p.npad = c->npad_fc; p.L = c->L; p.s2 = c->s2; p.n = c->F * c->npad_fc; d.push_back(p);
        p.kind = 2; p.src = params + c->poff[20]; p.src2 = nullptr; p.dst = c->dipack; p.npad = c->npad_di; p.n = c->F * c->npad_di; d.push_back(p);
        p.kind = 3; p.src = params + c->poff[38]; p.dst = c->wout_t; p.A = 32; p.n = 288; d.push_back(p);
        HIP_CHECK_RET(hipMemcpyAsync(c->d_descs, d.data(), d.size() * sizeof(PackDesc), hipMemcpyHostToDevice, st));
        c->packed_for = params;
    }
    ProfScope ps(c, "pack_weights", 0, 0, st);
    hipLaunchKernelGGL((pack_kernel<T>), dim3(128, (unsigned)d.size()), dim3(256), 0, st, c->d_descs);
    LAUNCH_CHECK("pack_kernel");
    return 0;
}

// ---- BatchNorm finalisation: folded into the consumer's prologue (BnFuse, common.cuh) or a standalone launch ----
static BnFuse make_fuse_fwd(vae_ctx* c, int i, const float* params, float* bn_running, int64_t* nbt) {
    const BnLayer& l = c->lay[i];
    BnFuse f; memset(&f, 0, sizeof(f));
    f.stat = l.stat_f; f.gamma = params + c->poff[l.p_gamma]; f.beta = params + c->poff[l.p_beta]; f.block = l.block;
    f.running_mean = bn_running ? bn_running + c->bnoff[i] : nullptr; f.running_var = bn_running ? bn_running + c->bnoff[i] + l.C : nullptr;
    f.nbt = nbt ? reinterpret_cast<long long*>(nbt) + i : nullptr;
    f.C = l.C; f.count = (double)c->B * l.H * l.W; f.eps = kBnEps; f.momentum = kBnMom; f.update_running = bn_running != nullptr;
    f.mode = BNF_FWD;
    return f;
}
static BnFuse make_fuse_bwd(vae_ctx* c, int i, const float* params, float* grads) {
    const BnLayer& l = c->lay[i];
    BnFuse f; memset(&f, 0, sizeof(f));
    f.stat = l.stat_b; f.gamma = params + c->poff[l.p_gamma]; f.block = l.block;
    f.dgamma = grads + c->poff[l.p_gamma]; f.dbeta = grads + c->poff[l.p_beta]; f.dconv_bias = grads + c->poff[l.p_convb];
    f.C = l.C; f.count = (double)c->B * l.H * l.W; f.mode = BNF_BWD;
    return f;
}
static int bn_finalize_now(vae_ctx* c, const BnFuse& f, hipStream_t st) {
    ProfScope ps(c, f.mode == BNF_FWD ? "bn_fwd_finalize" : "bn_bwd_finalize", 0, 0, st);
    hipLaunchKernelGGL(bn_finalize_kernel, dim3(1), dim3(256), 0, st, f);
    LAUNCH_CHECK("bn_finalize_kernel");
    return 0;
}
// Coefficients of layer i for the kernel that stages its tensor next.  Train mode with a fusing consumer: returns
// the BnFuse descriptor (mode BNF_FWD) and launches nothing; otherwise the block is filled by a standalone launch
// (batch statistics, or running statistics in eval mode) and the returned descriptor has mode BNF_NONE.
static int input_bn_fwd(vae_ctx* c, int i, const float* params, float* bn_running, int64_t* nbt, int train, bool consumer_fuses,
                        BnFuse* out, hipStream_t st) {
    memset(out, 0, sizeof(*out));
    const BnLayer& l = c->lay[i];
    if (train) {
        BnFuse f = make_fuse_fwd(c, i, params, bn_running, nbt);
        if (consumer_fuses && c->use_fused_bn) { *out = f; return 0; }
        return bn_finalize_now(c, f, st);
    }
    if (!bn_running) return vae_set_error("vae_forward", "eval mode needs running statistics");
    hipLaunchKernelGGL(bn_eval_coef_kernel, dim3(1), dim3(256), 0, st, params + c->poff[l.p_gamma], params + c->poff[l.p_beta],
                       bn_running + c->bnoff[i], bn_running + c->bnoff[i] + l.C, l.block, l.C, kBnEps);
    LAUNCH_CHECK("bn_eval_coef_kernel");
    return 0;
}
template <typename T> static bool will_pipe(vae_ctx* c, const ConvArgs<T>& a) { return c->use_pipelined && a.Cout <= c->knob_pipe_max_cout && fits_i32(a); }

// decoder half of the forward (models.py:147-175): decoder_input -> 3x ConvT blocks -> final_layer
// Weight gradients are consumed only by the optimiser: with use_side_stream they run on the context's side
// stream (own slab buffer), forked from the caller's stream at the point their inputs are ready, while the
// input-gradient chain - the critical path of the backward - continues on the caller's stream; the two are
// joined at the end of vae_backward.  fork_side returns the stream (and slab) the forked work should use.
struct SideFork { hipStream_t st; float* slab; int rc; };
static SideFork fork_side(vae_ctx* c, hipStream_t st, int which = -1) {
    SideFork f{st, c->slab, 0};
    if (!c->use_side_stream) return f;
    const int s = which >= 0 ? which : (c->side_rr++ % vae_ctx::NSIDE);
    hipEvent_t ev = c->ev_fork[c->fork_rr++ % vae_ctx::NFORK];
    if (hipEventRecord(ev, st) != hipSuccess || hipStreamWaitEvent(c->side[s], ev, 0) != hipSuccess) {
        f.rc = vae_set_error("fork_side", "event record/wait failed"); return f;
    }
    f.st = c->side[s]; f.slab = c->side_slab[s];
    return f;
}
// join every side stream into `st`
static int join_sides(vae_ctx* c, hipStream_t st) {
    if (!c->use_side_stream) return 0;
    for (int i = 0; i < vae_ctx::NSIDE; ++i) {
        HIP_CHECK_RET(hipEventRecord(c->ev_join[i], c->side[i]));
        HIP_CHECK_RET(hipStreamWaitEvent(st, c->ev_join[i], 0));
    }
    return 0;
}
// join the communication stream lent out by vae_comm_stream (work the caller enqueued on it, e.g. an all-reduce)
static int join_comm(vae_ctx* c, hipStream_t st) {
    if (!c->comm_busy) return 0;
    HIP_CHECK_RET(hipEventRecord(c->ev_comm, c->comm));
    HIP_CHECK_RET(hipStreamWaitEvent(st, c->ev_comm, 0));
    c->comm_busy = 0;
    return 0;
}
template <typename T>
static int decode_impl(vae_ctx* c, const float* z, int B, const float* params, float* bn_running, int64_t* nbt, int train,
                       const float* x, float* xhat, hipStream_t st) {
    const int H = c->H, L = c->L;
    static const char* kLayerTag[8] = {"encoder.0", "encoder.1", "encoder.2", "encoder.3", "decoder.0", "decoder.1", "decoder.2", "final_layer.0"};
    c->tag = "latent";
    // decoder_input
    {
        dim3 grid((unsigned)(c->F / 256), (B + 15) / 16);
        ProfScope ps(c, "decin_fwd", (double)sizeof(T) * B * (double)c->F + 4.0 * c->F * L, 2.0 * B * c->F * L, st);
        hipLaunchKernelGGL((decin_fwd_kernel<T>), grid, dim3(256), 16 * L * 4, st, z, params + c->poff[20], params + c->poff[21],
                           reinterpret_cast<T*>(c->d0), B, (int)c->F, L, c->s2);
        LAUNCH_CHECK("decin_fwd_kernel");
    }
    for (int i = 4; i < 8; ++i) {
        c->tag = kLayerTag[i];
        ConvArgs<T> a; memset(&a, 0, sizeof(a));
        if (i == 4) { a.src0 = reinterpret_cast<const T*>(c->d0); a.coef = c->ident; a.slope = 1.f; a.Cin = 256; }
        else { a.src0 = reinterpret_cast<const T*>(c->lay[i - 1].y); a.coef = c->lay[i - 1].block; a.slope = kSlope; a.Cin = c->lay[i - 1].C; }
        a.wp = reinterpret_cast<const T*>(c->wp_fwd[i]); a.bias = params + c->poff[c->lay[i].p_convb];
        a.out = reinterpret_cast<T*>(c->lay[i].y); a.stat = c->lay[i].stat_f;
        a.B = B; a.Hs = c->lay[i].H / 2; a.Ws = c->lay[i].W / 2; a.Cout = c->lay[i].C; a.epi = EPI_FWD;
        if (i > 4 && input_bn_fwd(c, i - 1, params, bn_running, nbt, train, will_pipe(c, a), &a.fuse, st)) return -1;
        if (launch_up<T>(c, a, st)) return -1;
    }
    // output conv + sigmoid + reconstruction loss/gradient
    c->tag = "final_layer.3";
    {
        ConvOutArgs a;
        a.yf = c->lay[7].y; a.coef = c->lay[7].block; a.wt = c->wout_t; a.bias = params + c->poff[39]; a.target = x;
        a.xhat = xhat; a.dlogit = c->dlogit; a.accum = c->accum; a.B = B; a.H = H; a.W = H;
        a.inv_n = (float)(1.0 / ((double)B * H * H)); a.slope = kSlope;
        const bool mfma_out = sizeof(T) == 2 && c->use_mfma_convout && 64.0 * B * H * H < 4294967296.0;   // 32-bit byte offsets
        BnFuse f7;
        if (input_bn_fwd(c, 7, params, bn_running, nbt, train, mfma_out, &f7, st)) return -1;
        ProfScope ps(c, "convout_fwd+bce", ((double)sizeof(T) * 32 + 12.0) * B * H * H, 2.0 * 9 * 32 * B * H * H, st);
        if (mfma_out) {
            ConvOutFwdMfmaArgs m; m.fuse = f7; m.rev = c->knob_rev & 1;
            m.yf = reinterpret_cast<const bf16*>(c->lay[7].y); m.coef = a.coef; m.wt = a.wt; m.bias = a.bias; m.target = x;
            m.xhat = xhat; m.dlogit = c->dlogit; m.accum = c->accum; m.B = B; m.H = H; m.W = H; m.n_tiles = B * (H / 8) * (H / 32);
            m.inv_n = a.inv_n; m.slope = kSlope;
            hipLaunchKernelGGL(convout_fwd_mfma_kernel, dim3(std::min(m.n_tiles, c->knob_convout_grid)), dim3(256), 0, st, m);
        } else {
            hipLaunchKernelGGL((convout_fwd_kernel<T>), dim3(B * (H / 16) * (H / 32)), dim3(256), 0, st, a);
        }
        LAUNCH_CHECK("convout_fwd_kernel");
    }
    return 0;
}

template <typename T>
static int forward_impl(vae_ctx* c, const float* x, int B, const float* params, float* bn_running, int64_t* nbt,
                        const float* eps, uint64_t seed, int train, float* xhat, float* mu, float* lv, float* z, hipStream_t st) {
    const int H = c->H, L = c->L;
    c->B = B; c->trained = train; c->x = x; c->xhat = xhat; c->mu = mu; c->lv = lv; c->z = z;
    HIP_CHECK_RET(hipMemsetAsync(c->dstats, 0, c->n_dstats * sizeof(double), st)); c->bwd_dirty = 0; c->walk_dir = 1;
    static const char* kLayerTag[8] = {"encoder.0", "encoder.1", "encoder.2", "encoder.3", "decoder.0", "decoder.1", "decoder.2", "final_layer.0"};
    // encoder block 0 (reads the raw f32 weights); the MFMA layers' packed weight images are built meanwhile
    c->tag = kLayerTag[0];
    {
        SideFork f = fork_side(c, st);
        if (f.rc) return f.rc;
        if (pack_weights<T>(c, params, f.st)) return -1;
        if (!eps && (c->knob_lean & 1)) {   // the reparameterisation noise is input-independent: drawn beside the first conv, not in the latent chain
            hipLaunchKernelGGL(counter_normal_kernel, dim3((B * L + 255) / 256), dim3(256), 0, f.st, c->eps, (long)B * L, (unsigned long long)seed, 5ULL);
            LAUNCH_CHECK("counter_normal_kernel");
        }
        if (c->use_side_stream) HIP_CHECK_RET(hipEventRecord(c->ev_pack, f.st));
    }
    {
        const long P = (long)B * (H / 2) * (H / 2);
        const int grid = (int)std::min<long>((P + 63) / 64, c->knob_conv1_grid);   // few workgroups: one f64 atomic per channel each
        ProfScope ps(c, "conv1_fwd", 4.0 * B * H * H + (double)sizeof(T) * 32.0 * P, 2.0 * 9 * 32 * P, st);
        hipLaunchKernelGGL((conv1_fwd_kernel<T>), dim3(grid), dim3(256), 0, st, x, params + c->poff[0], params + c->poff[1],
                           reinterpret_cast<T*>(c->lay[0].y), c->lay[0].stat_f, B, H, H);
        LAUNCH_CHECK("conv1_fwd_kernel");
        if (c->use_side_stream) HIP_CHECK_RET(hipStreamWaitEvent(st, c->ev_pack, 0));
    }
    for (int i = 1; i < 4; ++i) {
        c->tag = kLayerTag[i];
        ConvArgs<T> a; memset(&a, 0, sizeof(a));
        a.src0 = reinterpret_cast<const T*>(c->lay[i - 1].y); a.coef = c->lay[i - 1].block; a.slope = kSlope;
        a.wp = reinterpret_cast<const T*>(c->wp_fwd[i]); a.bias = params + c->poff[c->lay[i].p_convb];
        a.out = reinterpret_cast<T*>(c->lay[i].y); a.stat = c->lay[i].stat_f;
        a.B = B; a.Hs = c->lay[i].H; a.Ws = c->lay[i].W; a.Cin = c->lay[i - 1].C; a.Cout = c->lay[i].C; a.epi = EPI_FWD;
        if (input_bn_fwd(c, i - 1, params, bn_running, nbt, train, will_pipe(c, a), &a.fuse, st)) return -1;
        if (launch_down<T>(c, a, st)) return -1;
    }
    // fc_mu | fc_var, reparameterize
    c->tag = "latent";
    {
        DenseArgs<T> a; memset(&a, 0, sizeof(a));
        a.A = reinterpret_cast<const T*>(c->lay[3].y); a.coef = c->lay[3].block; a.slope = kSlope; a.C = 256;
        a.Bp = reinterpret_cast<const T*>(c->fcpack); a.M = B; a.K = (int)c->F; a.Npad = c->npad_fc;
        if (input_bn_fwd(c, 3, params, bn_running, nbt, train, true, &a.fuse, st)) return -1;
        int nsplit;
        if (launch_dense<T>(c, a, &nsplit, st)) return -1;
        if (eps) HIP_CHECK_RET(hipMemcpyAsync(c->eps, eps, (size_t)B * L * 4, hipMemcpyDeviceToDevice, st));
        else if (!(c->knob_lean & 1)) {
            hipLaunchKernelGGL(counter_normal_kernel, dim3((B * L + 255) / 256), dim3(256), 0, st, c->eps, (long)B * L, (unsigned long long)seed, 5ULL);
            LAUNCH_CHECK("counter_normal_kernel");
        }
        LatentFwdArgs la;
        la.slab = c->slab; la.nslab = nsplit; la.npad = c->npad_fc; la.bmu = params + c->poff[17]; la.bvar = params + c->poff[19];
        la.eps = c->eps; la.mu = mu; la.lv = lv; la.z = z; la.accum = c->accum; la.B = B; la.L = L;
        hipLaunchKernelGGL(latent_fwd_kernel, dim3((B * L * LAT_LANES + 255) / 256), dim3(256), 0, st, la);
        LAUNCH_CHECK("latent_fwd_kernel");
    }
    return decode_impl<T>(c, z, B, params, bn_running, nbt, train, x, xhat, st);
}


template <typename T>
static int wgrad_on_side(vae_ctx* c, WgradArgs<T> w, float* dw_out, hipStream_t st) {
    SideFork f = fork_side(c, st);
    if (f.rc) return f.rc;
    return launch_wgrad<T>(c, w, dw_out, f.st, f.slab);
}

template <typename T>
static int backward_first(vae_ctx* c, const float* x, const float* params, float* grads, const float* g_xhat, const float* gscale,
                         const float* g_mu, const float* g_lv, const float* g_z, const float* g_pre, float kld_weight, int add_kl,
                         hipStream_t st) {
    if (!c->B || !c->trained) return vae_set_error("vae_backward", "no train-mode forward to differentiate");
    const int B = c->B, H = c->H, L = c->L;
    size_t nfwd = 0;
    for (int i = 0; i < 8; ++i) nfwd += 2 * kBnC[i] * STAT_R;
    if (c->bwd_dirty) {   // (the forward zeroed every accumulator; only a repeated backward has to clear its own)
        HIP_CHECK_RET(hipMemsetAsync(c->dstats + nfwd, 0, nfwd * sizeof(double), st));   // stat_b
        for (int rep = 0; rep < STAT_R; ++rep) HIP_CHECK_RET(hipMemsetAsync(c->accum + rep * 8 + 2, 0, sizeof(double), st));
    }
    c->bwd_dirty = 1;
    const float* dl_src = c->dlogit; const float* dl_scale = gscale;
    if (g_xhat || !add_kl) {
        // explicit upstream gradient on xhat (plus, when add_kl, the fused standard-ELBO term)
        const long n = (long)B * H * H;
        hipLaunchKernelGGL(dlogit_combine_kernel, dim3((unsigned)std::min<long>((n + 255) / 256, 4096)), dim3(256), 0, st,
                           g_xhat, c->xhat, add_kl ? c->dlogit : nullptr, gscale, c->dlogit2, n);
        LAUNCH_CHECK("dlogit_combine_kernel");
        dl_src = c->dlogit2; dl_scale = nullptr;
    }
    static const char* kLayerTag[8] = {"encoder.0", "encoder.1", "encoder.2", "encoder.3", "decoder.0", "decoder.1", "decoder.2", "final_layer.0"};
    // output conv backward + final_layer BN/LeakyReLU prologue
    c->tag = "final_layer.3";
    int cgrid = 0;
    {
        ConvOutBwdArgs a;
        a.yf = c->lay[7].y; a.ocoef = c->lay[7].block; a.wt = c->wout_t; a.dlogit = dl_src; a.gscale = dl_scale;
        a.dz = c->lay[7].dz; a.slab = c->use_side_stream ? c->side_slab[0] : c->slab; a.stat = c->lay[7].stat_b; a.dbias = c->accum + 2; a.B = B; a.H = H; a.W = H; a.slope = kSlope;
        const long P = (long)B * H * H;
        int grid = (int)std::min<long>((P + 63) / 64, 1024);
        ProfScope ps(c, "convout_bwd(dgrad+wgrad+bn prologue)", ((double)sizeof(T) * 64 + 4.0) * P, 3.0 * 2 * 9 * 32 * P, st);
        if (sizeof(T) == 2 && c->use_mfma_convout) {
            ConvOutBwdMfmaArgs m; m.rev = (c->knob_rev >> 1) & 1;
            m.yf = reinterpret_cast<const bf16*>(c->lay[7].y); m.ocoef = a.ocoef; m.wt = a.wt; m.dlogit = a.dlogit; m.gscale = a.gscale;
            m.dz = reinterpret_cast<bf16*>(c->lay[7].dz); m.slab = a.slab; m.stat = a.stat; m.dbias = a.dbias;
            m.B = B; m.H = H; m.W = H; m.n_tiles = B * (H / 8) * (H / 32); m.slope = kSlope;
            grid = std::min(m.n_tiles, c->knob_convout_bwd_grid);
            hipLaunchKernelGGL(convout_bwd_mfma_kernel, dim3(grid), dim3(256), 0, st, m);
        } else {
            hipLaunchKernelGGL((convout_bwd_kernel<T>), dim3(grid), dim3(256), 0, st, a);
        }
        cgrid = grid;
        LAUNCH_CHECK("convout_bwd_kernel");
    }
    {
        SideFork f = fork_side(c, st, 0);   // the kernel above wrote its partial sums into side stream 0's slab
        if (f.rc) return f.rc;
        if (launch_reduce(f.slab, cgrid, 288, grads + c->poff[38], 1, 32, f.st, c)) return -1;
        hipLaunchKernelGGL(accum_to_f32_kernel, dim3(1), dim3(64), 0, f.st, c->accum + 2, grads + c->poff[39]);
        LAUNCH_CHECK("accum_to_f32_kernel");
    }
    // decoder stack: ConvTranspose2d layers 7 (final_layer.0), 6, 5, 4
    for (int i = 7; i >= 4; --i) {
        c->tag = kLayerTag[i];
        const BnLayer& l = c->lay[i];
        const int Cin = i == 4 ? 256 : c->lay[i - 1].C;
        WgradArgs<T> w; memset(&w, 0, sizeof(w));
        if (i == 4) { w.s0 = reinterpret_cast<const T*>(c->d0); w.scoef = c->ident; w.sslope = 1.f; }
        else { w.s0 = reinterpret_cast<const T*>(c->lay[i - 1].y); w.scoef = c->lay[i - 1].block; w.sslope = kSlope; }
        w.s_two = 0;
        w.g0 = reinterpret_cast<const T*>(l.dz); w.g1 = reinterpret_cast<const T*>(l.y); w.gcoef = l.block + LC_P0 * l.C; w.gslope = 1.f; w.g_two = 1;
        w.B = B; w.Hs = l.H / 2; w.Ws = l.W / 2; w.CA = Cin; w.CB = l.C;
        ConvArgs<T> a; memset(&a, 0, sizeof(a));
        a.src0 = reinterpret_cast<const T*>(l.dz); a.src1 = reinterpret_cast<const T*>(l.y); a.coef = l.block + LC_P0 * l.C; a.slope = 1.f; a.two_src = 1;
        a.wp = reinterpret_cast<const T*>(c->wp_dg[i]);
        a.B = B; a.Hs = l.H / 2; a.Ws = l.W / 2; a.Cin = l.C; a.Cout = Cin;
        if (i == 4) { a.out = reinterpret_cast<T*>(c->dd0); a.epi = EPI_PLAIN; }
        else {
            a.out = reinterpret_cast<T*>(c->lay[i - 1].dz); a.yout = reinterpret_cast<const T*>(c->lay[i - 1].y);
            a.ocoef = c->lay[i - 1].block; a.oslope = kSlope; a.stat = c->lay[i - 1].stat_b; a.epi = EPI_BWD;
        }
        // BatchNorm backward of this layer: folded into both consumers (the input-gradient kernel records it)
        BnFuse fb = make_fuse_bwd(c, i, params, grads);
        if (!(c->use_fused_bn && will_pipe(c, a))) { if (bn_finalize_now(c, fb, st)) return -1; fb.mode = BNF_NONE; }
        w.fuse = fb; a.fuse = fb;
        if (wgrad_on_side<T>(c, w, grads + c->poff[l.p_convw], st)) return -1;
        if (launch_down<T>(c, a, st)) return -1;
    }
    return 0;
}

// second half of the backward: decoder_input / latent / fc / encoder (everything below the decoder stack)
template <typename T>
static int backward_second(vae_ctx* c, const float* x, const float* params, float* grads, const float* gscale,
                           const float* g_mu, const float* g_lv, const float* g_z, const float* g_pre, float kld_weight, int add_kl,
                           hipStream_t st) {
    const int B = c->B, H = c->H, L = c->L;
    static const char* kLayerTag[8] = {"encoder.0", "encoder.1", "encoder.2", "encoder.3", "decoder.0", "decoder.1", "decoder.2", "final_layer.0"};
    // decoder_input backward, reparameterisation + KL backward
    c->tag = "latent";
    {
        {
            // batch split over grid.z (8 slices) -> slabs -> one reduce per tensor
            const int nz = std::max(1, std::min(8, B / 8)), bsplit = (B + nz - 1) / nz;
            SideFork f = fork_side(c, st);
            if (f.rc) return f.rc;
            float* sw = f.slab; float* sb = f.slab + (size_t)nz * c->F * L;
            dim3 grid((unsigned)(c->F / 256), (L + 31) / 32, nz);
            {
                ProfScope ps(c, "decin_wgrad", (double)sizeof(T) * B * (double)c->F + 4.0 * c->F * L, 2.0 * B * c->F * L, f.st);
                hipLaunchKernelGGL((decin_wgrad_kernel<T>), grid, dim3(256), 0, f.st, reinterpret_cast<const T*>(c->dd0), c->z, sw, sb, B, (int)c->F, L, c->s2, bsplit);
                LAUNCH_CHECK("decin_wgrad_kernel");
            }
            if (launch_reduce(sw, nz, (size_t)c->F * L, grads + c->poff[20], 0, 0, f.st, c)) return -1;
            if (launch_reduce(sb, nz, (size_t)c->F, grads + c->poff[21], 0, 0, f.st, c)) return -1;
        }
        DenseArgs<T> a; memset(&a, 0, sizeof(a));
        a.A = reinterpret_cast<const T*>(c->dd0); a.coef = nullptr; a.slope = 1.f; a.C = 256;
        a.Bp = reinterpret_cast<const T*>(c->dipack); a.M = B; a.K = (int)c->F; a.Npad = c->npad_di;
        int nsplit;
        if (launch_dense<T>(c, a, &nsplit, st)) return -1;
        LatentBwdArgs lb;
        lb.slab = c->slab; lb.nslab = nsplit; lb.npad = c->npad_di; lb.mu = c->mu; lb.lv = c->lv; lb.eps = c->eps; lb.gscale = gscale;
        lb.gmu = g_mu; lb.glv = g_lv; lb.gz = g_z; lb.dlat = c->dlat; lb.B = B; lb.L = L; lb.kld_weight = kld_weight; lb.add_kl = add_kl;
        hipLaunchKernelGGL(latent_bwd_kernel, dim3((B * L * LAT_LANES + 255) / 256), dim3(256), 0, st, lb);
        LAUNCH_CHECK("latent_bwd_kernel");
        SideFork f = fork_side(c, st);
        if (f.rc) return f.rc;
        hipLaunchKernelGGL(colsum_kernel, dim3(2 * L), dim3(64), 0, f.st, c->dlat, B, 2 * L, grads + c->poff[17], grads + c->poff[19], L);
        LAUNCH_CHECK("colsum_kernel");
    }
    // fc_mu / fc_var backward
    {
        FcWgradArgs<T> w;
        w.dlat = c->dlat; w.y = reinterpret_cast<const T*>(c->lay[3].y); w.coef = c->lay[3].block; w.slope = kSlope;
        w.dwmu = grads + c->poff[16]; w.dwvar = grads + c->poff[18]; w.B = B; w.F = (int)c->F; w.L = L; w.s2 = c->s2;
        {
            const int nz = std::max(1, std::min(8, B / 8));
            w.bsplit = (B + nz - 1) / nz;
            SideFork f = fork_side(c, st);   // (no new dependency: the side stream is already past latent_bwd)
            if (f.rc) return f.rc;
            float* smu = f.slab; float* svar = f.slab + (size_t)nz * L * c->F;
            w.dwmu = smu; w.dwvar = svar;
            {
                ProfScope ps(c, "fc_wgrad", (double)sizeof(T) * B * (double)c->F + 8.0 * c->F * L, 4.0 * B * c->F * L, f.st);
                hipLaunchKernelGGL((fc_wgrad_kernel<T>), dim3((unsigned)(c->F / 256), (2 * L + 31) / 32, nz), dim3(256), 0, f.st, w);
                LAUNCH_CHECK("fc_wgrad_kernel");
            }
            if (launch_reduce(smu, nz, (size_t)L * c->F, grads + c->poff[16], 0, 0, f.st, c)) return -1;
            if (launch_reduce(svar, nz, (size_t)L * c->F, grads + c->poff[18], 0, 0, f.st, c)) return -1;
        }
        FcDgradArgs<T> d;
        d.dlat = c->dlat; d.wp = reinterpret_cast<const T*>(c->fcpack); d.npad = c->npad_fc; d.y = reinterpret_cast<const T*>(c->lay[3].y);
        d.ocoef = c->lay[3].block; d.slope = kSlope; d.gpre = g_pre; d.dz = reinterpret_cast<T*>(c->lay[3].dz); d.stat = c->lay[3].stat_b;
        d.B = B; d.F = (int)c->F; d.L2 = 2 * L; d.s2 = c->s2;
        ProfScope ps2(c, "fc_dgrad", (double)sizeof(T) * (2.0 * B * c->F + 2.0 * c->F * L), 4.0 * B * c->F * L, st);
        d.bt_per_wg = std::max(16, ((B + 7) / 8 + 15) / 16 * 16);   // <= 8 workgroups per channel: fewer same-address atomics
        hipLaunchKernelGGL((fc_dgrad_kernel<T>), dim3((unsigned)(c->F / 256), (B + d.bt_per_wg - 1) / d.bt_per_wg), dim3(256), 2 * L * 16 * 4, st, d);
        LAUNCH_CHECK("fc_dgrad_kernel");
    }
    // encoder stack: Conv2d layers 3, 2, 1 on MFMA, then block 0
    for (int i = 3; i >= 1; --i) {
        c->tag = kLayerTag[i];
        const BnLayer& l = c->lay[i]; const BnLayer& lp = c->lay[i - 1];
        WgradArgs<T> w; memset(&w, 0, sizeof(w));
        w.s0 = reinterpret_cast<const T*>(l.dz); w.s1 = reinterpret_cast<const T*>(l.y); w.scoef = l.block + LC_P0 * l.C; w.sslope = 1.f; w.s_two = 1;
        w.g0 = reinterpret_cast<const T*>(lp.y); w.gcoef = lp.block; w.gslope = kSlope; w.g_two = 0;
        w.B = B; w.Hs = l.H; w.Ws = l.W; w.CA = l.C; w.CB = lp.C;
        ConvArgs<T> a; memset(&a, 0, sizeof(a));
        a.src0 = reinterpret_cast<const T*>(l.dz); a.src1 = reinterpret_cast<const T*>(l.y); a.coef = l.block + LC_P0 * l.C; a.slope = 1.f; a.two_src = 1;
        a.wp = reinterpret_cast<const T*>(c->wp_dg[i]);
        a.out = reinterpret_cast<T*>(lp.dz); a.yout = reinterpret_cast<const T*>(lp.y); a.ocoef = lp.block; a.oslope = kSlope; a.stat = lp.stat_b; a.epi = EPI_BWD;
        a.B = B; a.Hs = l.H; a.Ws = l.W; a.Cin = l.C; a.Cout = lp.C;
        BnFuse fb = make_fuse_bwd(c, i, params, grads);
        if (!(c->use_fused_bn && will_pipe(c, a))) { if (bn_finalize_now(c, fb, st)) return -1; fb.mode = BNF_NONE; }
        w.fuse = fb; a.fuse = fb;
        if (wgrad_on_side<T>(c, w, grads + c->poff[l.p_convw], st)) return -1;
        if (launch_up<T>(c, a, st)) return -1;
    }
    {
        c->tag = kLayerTag[0];
        BnFuse fb0 = make_fuse_bwd(c, 0, params, grads);
        if (!c->use_fused_bn || !(c->knob_lean & 2)) { if (bn_finalize_now(c, fb0, st)) return -1; fb0.mode = BNF_NONE; }
        const long P = (long)B * (H / 2) * (H / 2);
        const int grid = (int)std::min<long>((P + 63) / 64, 512);
        // last link of the chain: stays on the caller's stream (a side stream would only add an event round trip)
        SideFork f{st, c->slab, 0};
        {
            ProfScope ps(c, "conv1_wgrad", 4.0 * B * H * H + (double)sizeof(T) * 64.0 * P, 2.0 * 9 * 32 * P, f.st);
            hipLaunchKernelGGL((conv1_wgrad_kernel<T>), dim3(grid), dim3(256), 0, f.st, x, reinterpret_cast<const T*>(c->lay[0].dz),
                               reinterpret_cast<const T*>(c->lay[0].y), c->lay[0].block + LC_P0 * 32, f.slab, B, H, H, fb0);
            LAUNCH_CHECK("conv1_wgrad_kernel");
        }
        if (launch_reduce(f.slab, grid, 288, grads + c->poff[0], 32, 1, f.st, c)) return -1;
    }
    return join_sides(c, st);
}

// part 0: whole backward; 1: output conv + decoder stack, ending with every decoder gradient complete on `st`
// (data-parallel callers start that bucket's all-reduce here); 2: the rest.
template <typename T>
static int backward_impl(vae_ctx* c, const float* x, const float* params, float* grads, const float* g_xhat, const float* gscale,
                         const float* g_mu, const float* g_lv, const float* g_z, const float* g_pre, float kld_weight, int add_kl,
                         int part, hipStream_t st) {
    if (part < 0 || part > 2) return vae_set_error("vae_backward", "part must be 0, 1 or 2");
    if (part != 2) {
        if (backward_first<T>(c, x, params, grads, g_xhat, gscale, g_mu, g_lv, g_z, g_pre, kld_weight, add_kl, st)) return -1;
        c->bwd_half_done = 1;
        if (part == 1) return join_sides(c, st);
    } else if (!c->bwd_half_done) return vae_set_error("vae_backward", "part 2 before part 1");
    c->bwd_half_done = 0;
    if (backward_second<T>(c, x, params, grads, gscale, g_mu, g_lv, g_z, g_pre, kld_weight, add_kl, st)) return -1;
    return join_comm(c, st);
}

// ---------------------------------------------------------------------------
extern "C" int vae_forward(vae_ctx* c, const float* x, int B, const float* params, float* bn_running, int64_t* nbt,
                           const float* eps, uint64_t seed, int train, float* xhat, float* mu, float* lv, float* z, vae_stream_t stream) {
    if (!c) return vae_set_error("vae_forward", "null ctx");
    if (B < 1 || B > c->maxB) return vae_set_error("vae_forward", "batch exceeds the context's max_batch");
    if (!x || !params || !xhat || !mu || !lv || !z) return vae_set_error("vae_forward", "null tensor pointer");
    hipStream_t st = (hipStream_t)stream;
    return c->dtype == VAE_DTYPE_BF16 ? forward_impl<bf16>(c, x, B, params, bn_running, nbt, eps, seed, train, xhat, mu, lv, z, st)
                                      : forward_impl<float>(c, x, B, params, bn_running, nbt, eps, seed, train, xhat, mu, lv, z, st);
}

extern "C" int vae_decode(vae_ctx* c, const float* z, int B, const float* params, float* bn_running, int64_t* nbt, int train,
                          float* xhat, vae_stream_t stream) {
    if (!c) return vae_set_error("vae_decode", "null ctx");
    if (B < 1 || B > c->maxB) return vae_set_error("vae_decode", "batch exceeds the context's max_batch");
    if (!z || !params || !xhat) return vae_set_error("vae_decode", "null tensor pointer");
    hipStream_t st = (hipStream_t)stream;
    c->B = B; c->trained = 0;   // a decode-only pass cannot be differentiated
    HIP_CHECK_RET(hipMemsetAsync(c->dstats, 0, c->n_dstats * sizeof(double), st)); c->bwd_dirty = 0;
    int rc = c->dtype == VAE_DTYPE_BF16 ? pack_weights<bf16>(c, params, st) : pack_weights<float>(c, params, st);
    if (rc) return rc;
    // the reconstruction-loss side outputs of the output-conv kernel are unused here: xhat doubles as the target
    return c->dtype == VAE_DTYPE_BF16 ? decode_impl<bf16>(c, z, B, params, bn_running, nbt, train, xhat, xhat, st)
                                      : decode_impl<float>(c, z, B, params, bn_running, nbt, train, xhat, xhat, st);
}

extern "C" int vae_loss(vae_ctx* c, float kld_weight, float* out3, vae_stream_t stream) {
    if (!c || !c->B) return vae_set_error("vae_loss", "no forward");
    hipLaunchKernelGGL(loss_finalize_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, c->accum, out3,
                       1.0 / ((double)c->B * c->H * c->H), 1.0 / (double)c->B, kld_weight, STAT_R);
    LAUNCH_CHECK("loss_finalize_kernel");
    return 0;
}

// Same scalars, computed beside the backward instead of in front of it: enqueued on one of the context's side streams
// (ordered after `stream`), so out3 is ordered into the caller's stream by the NEXT vae_backward / vae_backward_part
// on this context - for callers that only read the ELBO after the backward (train.py:644-674 reads it after the step).
extern "C" int vae_loss_deferred(vae_ctx* c, float kld_weight, float* out3, vae_stream_t stream) {
    if (!c || !c->B) return vae_set_error("vae_loss_deferred", "no forward");
    if (!c->trained) return vae_set_error("vae_loss_deferred", "needs a train-mode forward (a backward must follow)");
    SideFork f = (c->knob_lean & 4) ? fork_side(c, (hipStream_t)stream) : SideFork{(hipStream_t)stream, c->slab, 0};
    if (f.rc) return f.rc;
    hipLaunchKernelGGL(loss_finalize_kernel, dim3(1), dim3(64), 0, f.st, c->accum, out3,
                       1.0 / ((double)c->B * c->H * c->H), 1.0 / (double)c->B, kld_weight, STAT_R);
    LAUNCH_CHECK("loss_finalize_kernel");
    return 0;
}

static double* g_generic_accum = nullptr;
extern "C" int vae_elbo_generic(const float* xhat, const float* target, const float* mu, const float* lv, int64_t n, int B, int L,
                                float kld_weight, float* out3, float* g_xhat, float* g_mu, float* g_lv, vae_stream_t stream) {
    hipStream_t st = (hipStream_t)stream;
    if (!g_generic_accum) HIP_CHECK_RET(hipMalloc(&g_generic_accum, 4 * sizeof(double)));
    HIP_CHECK_RET(hipMemsetAsync(g_generic_accum, 0, 4 * sizeof(double), st));
    hipLaunchKernelGGL(bce_kernel, dim3((unsigned)std::min<long>((n + 255) / 256, 2048)), dim3(256), 0, st, xhat, target, g_xhat, g_generic_accum, (long)n, (float)(1.0 / (double)n));
    LAUNCH_CHECK("bce_kernel");
    hipLaunchKernelGGL(kld_only_kernel, dim3((B * L + 255) / 256), dim3(256), 0, st, mu, lv, g_generic_accum, B * L, kld_weight / (float)B, g_mu, g_lv);
    LAUNCH_CHECK("kld_only_kernel");
    hipLaunchKernelGGL(loss_finalize_kernel, dim3(1), dim3(64), 0, st, g_generic_accum, out3, 1.0 / (double)n, 1.0 / (double)B, kld_weight, 1);
    LAUNCH_CHECK("loss_finalize_kernel");
    return 0;
}

// A non-blocking stream owned by the context, ordered after everything enqueued on `stream` so far.  Work the caller
// puts on it (the all-reduce of the decoder gradients after vae_backward_part(..., 1, ...)) is joined back into the
// caller's stream at the end of vae_backward_part(..., 2, ...).
extern "C" int vae_comm_stream(vae_ctx* c, vae_stream_t stream, vae_stream_t* out) {
    if (!c || !out) return vae_set_error("vae_comm_stream", "null argument");
    hipStream_t st = (hipStream_t)stream;
    hipEvent_t ev = c->ev_fork[c->fork_rr++ % vae_ctx::NFORK];
    HIP_CHECK_RET(hipEventRecord(ev, st));
    HIP_CHECK_RET(hipStreamWaitEvent(c->comm, ev, 0));
    c->comm_busy = 1;
    *out = (vae_stream_t)c->comm;
    return 0;
}
extern "C" int vae_backward_part(vae_ctx* c, const float* x, const float* params, float* grads, const float* g_xhat, const float* gscale,
                                 const float* g_mu, const float* g_lv, const float* g_z, const float* g_pre, float kld_weight, int add_kl,
                                 int part, vae_stream_t stream) {
    if (!c) return vae_set_error("vae_backward", "null ctx");
    if (!x || !params || !grads) return vae_set_error("vae_backward", "null tensor pointer");
    hipStream_t st = (hipStream_t)stream;
    return c->dtype == VAE_DTYPE_BF16 ? backward_impl<bf16>(c, x, params, grads, g_xhat, gscale, g_mu, g_lv, g_z, g_pre, kld_weight, add_kl, part, st)
                                      : backward_impl<float>(c, x, params, grads, g_xhat, gscale, g_mu, g_lv, g_z, g_pre, kld_weight, add_kl, part, st);
}
extern "C" int vae_backward(vae_ctx* c, const float* x, const float* params, float* grads, const float* g_xhat, const float* gscale,
                            const float* g_mu, const float* g_lv, const float* g_z, const float* g_pre, float kld_weight, int add_kl,
                            vae_stream_t stream) {
    return vae_backward_part(c, x, params, grads, g_xhat, gscale, g_mu, g_lv, g_z, g_pre, kld_weight, add_kl, 0, stream);
}

extern "C" int vae_adamw_step(float* params, const float* grads, float* m, float* v, int ngroups, const int64_t* offsets,
                              const int64_t* sizes, const float* lrs, const float* beta1s, float beta2, float eps, float weight_decay,
                              float grad_scale, int step, vae_stream_t stream) {
    if (ngroups < 1 || ngroups > 2) return vae_set_error("vae_adamw_step", "1 or 2 groups");
    if (step < 1) return vae_set_error("vae_adamw_step", "step is 1-based");
    AdamArgs a;
    a.p = params; a.g = grads; a.m = m; a.v = v; a.ngrp = ngroups; a.beta2 = beta2; a.eps = eps; a.weight_decay = weight_decay;
    a.grad_scale = grad_scale; a.step = step;
    long nmax = 0;
    for (int i = 0; i < ngroups; ++i) { a.grp[i].off = offsets[i]; a.grp[i].n = sizes[i]; a.grp[i].lr = lrs[i]; a.grp[i].beta1 = beta1s[i]; nmax = std::max<long>(nmax, sizes[i]); }
    hipLaunchKernelGGL(adamw_kernel, dim3((unsigned)std::min<long>((nmax + 255) / 256, 2048), ngroups), dim3(256), 0, (hipStream_t)stream, a);
    LAUNCH_CHECK("adamw_kernel");
    return 0;
}

extern "C" int vae_train_step(vae_ctx* c, const float* x, int B, float* params, float* grads, float* m, float* v, float* bn_running,
                              int64_t* nbt, const float* eps, uint64_t seed, float kld_weight, int ngroups, const int64_t* offsets,
                              const int64_t* sizes, const float* lrs, const float* beta1s, float beta2, float adam_eps,
                              float weight_decay, int step, float* xhat, float* mu, float* lv, float* z, float* out3, vae_stream_t stream) {
    if (vae_forward(c, x, B, params, bn_running, nbt, eps, seed, 1, xhat, mu, lv, z, stream)) return -1;
    if (vae_loss_deferred(c, kld_weight, out3, stream)) return -1;
    if (vae_backward(c, x, params, grads, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, kld_weight, 1, stream)) return -1;
    if (ngroups > 0 && vae_adamw_step(params, grads, m, v, ngroups, offsets, sizes, lrs, beta1s, beta2, adam_eps, weight_decay, 1.f, step, stream)) return -1;
    return 0;
}

// diagnostic: phase stamps of the pipelined down kernel for one layer ("encoder.3", epi) into out[grid*4*6]
extern "C" int vae_debug_stamps(vae_ctx* c, const char* tag, int epi, long long* out) {
    if (!c) return -1;
    c->dbg_buf = out; c->dbg_epi = epi; strncpy(c->dbg_tag, tag ? tag : "", sizeof(c->dbg_tag) - 1);
    return 0;
}

extern "C" int vae_profile(vae_ctx* c, int enable) {
    if (!c) return vae_set_error("vae_profile", "null ctx");
    for (auto& r : c->prof_recs) { (void)hipEventDestroy(r.e0); (void)hipEventDestroy(r.e1); }
    c->prof_recs.clear(); c->prof = enable;
    return 0;
}
// JSON: [{"name":..,"calls":n,"ms":total,"bytes":total algorithmic bytes,"flops":total}, ...]
extern "C" int vae_profile_report(vae_ctx* c, char* buf, int64_t cap) {
    if (!c) return vae_set_error("vae_profile_report", "null ctx");
    HIP_CHECK_RET(hipDeviceSynchronize());
    struct Agg { std::string name; int calls; double ms, bytes, flops; };
    std::vector<Agg> agg;
    for (auto& r : c->prof_recs) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, r.e0, r.e1) != hipSuccess) continue;
        Agg* a = nullptr;
        for (auto& x : agg) if (x.name == r.name) a = &x;
        if (!a) { agg.push_back({r.name, 0, 0, 0, 0}); a = &agg.back(); }
        a->calls += 1; a->ms += ms; a->bytes += r.bytes; a->flops += r.flops;
    }
    std::string out = "[";
    for (size_t i = 0; i < agg.size(); ++i) {
        char line[512];
        snprintf(line, sizeof(line), "%s{\"name\":\"%s\",\"calls\":%d,\"ms\":%.6f,\"bytes\":%.1f,\"flops\":%.1f}", i ? "," : "",
                 agg[i].name.c_str(), agg[i].calls, agg[i].ms, agg[i].bytes, agg[i].flops);
        out += line;
    }
    out += "]";
    if ((int64_t)out.size() + 1 > cap) return vae_set_error("vae_profile_report", "buffer too small");
    memcpy(buf, out.c_str(), out.size() + 1);
    return 0;
}

// JSON array of the profiled launch labels in launch order (for matching rocprofv3 dispatches to labels)
extern "C" int vae_profile_sequence(vae_ctx* c, char* buf, int64_t cap) {
    if (!c) return vae_set_error("vae_profile_sequence", "null ctx");
    std::string out = "[";
    for (size_t i = 0; i < c->prof_recs.size(); ++i) out += std::string(i ? "," : "") + "\"" + c->prof_recs[i].name + "\"";
    out += "]";
    if ((int64_t)out.size() + 1 > cap) return vae_set_error("vae_profile_sequence", "buffer too small");
    memcpy(buf, out.c_str(), out.size() + 1);
    return 0;
}

// JSON: [[name, start_ms, end_ms, algorithmic_bytes], ...] relative to the first recorded launch (shows the overlap of the streams)
extern "C" int vae_profile_timeline(vae_ctx* c, char* buf, int64_t cap) {
    if (!c) return vae_set_error("vae_profile_timeline", "null ctx");
    HIP_CHECK_RET(hipDeviceSynchronize());
    std::string out = "[";
    for (size_t i = 0; i < c->prof_recs.size(); ++i) {
        float t0 = 0.f, t1 = 0.f;
        (void)hipEventElapsedTime(&t0, c->prof_recs[0].e0, c->prof_recs[i].e0);
        (void)hipEventElapsedTime(&t1, c->prof_recs[0].e0, c->prof_recs[i].e1);
        char tmp[96]; snprintf(tmp, sizeof(tmp), "\",%.4f,%.4f,%.0f]", t0, t1, c->prof_recs[i].bytes);
        out += std::string(i ? ",[\"" : "[\"") + c->prof_recs[i].name + tmp;
    }
    out += "]";
    if ((int64_t)out.size() + 1 > cap) return vae_set_error("vae_profile_timeline", "buffer too small");
    memcpy(buf, out.c_str(), out.size() + 1);
    return 0;
}

extern "C" int vae_pre_latents(vae_ctx* c, float* out, vae_stream_t stream) {
    if (!c || !c->B) return vae_set_error("vae_pre_latents", "no forward");
    const long n = (long)c->B * c->F;
    if (c->dtype == VAE_DTYPE_BF16)
        hipLaunchKernelGGL((pre_latents_kernel<bf16>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                           reinterpret_cast<const bf16*>(c->lay[3].y), c->lay[3].block, kSlope, out, c->B, (int)c->F, c->s2);
    else
        hipLaunchKernelGGL((pre_latents_kernel<float>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                           reinterpret_cast<const float*>(c->lay[3].y), c->lay[3].block, kSlope, out, c->B, (int)c->F, c->s2);
    LAUNCH_CHECK("pre_latents_kernel");
    return 0;
}
extern "C" int vae_last_eps(vae_ctx* c, float* out, vae_stream_t stream) {
    if (!c || !c->B) return vae_set_error("vae_last_eps", "no forward");
    HIP_CHECK_RET(hipMemcpyAsync(out, c->eps, (size_t)c->B * c->L * 4, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return 0;
}

template <typename T>
__global__ void nhwc_to_nchw_kernel(const T* in, float* out, long n, int C, int HW) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int ch = i % C; const long pix = (i / C) % HW; const long b = i / ((long)C * HW);
    out[(b * C + ch) * HW + pix] = tofloat(in[i]);
}
extern "C" int vae_debug_tensor(vae_ctx* c, int which, float* out, int64_t capacity, vae_stream_t stream) {
    if (!c || !c->B) return vae_set_error("vae_debug_tensor", "no forward");
    const void* src; int C, HW;
    if (which >= 0 && which < 16) { const BnLayer& l = c->lay[which & 7]; src = which < 8 ? l.y : l.dz; C = l.C; HW = l.H * l.W; }
    else if (which == 16 || which == 17) { src = which == 16 ? c->d0 : c->dd0; C = 256; HW = c->s2; }
    else return vae_set_error("vae_debug_tensor", "bad tensor id");
    const long n = (long)c->B * C * HW;
    if (n > capacity) return vae_set_error("vae_debug_tensor", "output too small");
    if (c->dtype == VAE_DTYPE_BF16)
        hipLaunchKernelGGL((nhwc_to_nchw_kernel<bf16>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, reinterpret_cast<const bf16*>(src), out, n, C, HW);
    else
        hipLaunchKernelGGL((nhwc_to_nchw_kernel<float>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, reinterpret_cast<const float*>(src), out, n, C, HW);
    LAUNCH_CHECK("nhwc_to_nchw_kernel");
    return 0;
}

// Self-test of ds_read_b64_tr_b16: a 16x32 tile of 16-bit words M[k][c] = k*32 + c staged as
// [k][c]; the k-major fragment of lane (r,h) must come back as M[8h+j][r].
__global__ void selftest_tr16_kernel(int* bad) {
    __shared__ __attribute__((aligned(16))) short tile[16 * 32];
    const int lane = threadIdx.x;
    for (int i = lane; i < 16 * 32; i += 64) tile[i] = (short)i;
    __syncthreads();
    const int g4 = lane >> 4, li = lane & 15, q = li >> 2, p = li & 3, r = lane & 31, h = lane >> 5;
    int nbad = 0;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        const int k = 8 * (g4 >> 1) + 4 * half + q;
        const char* ad = reinterpret_cast<const char*>(tile) + k * 64 + (16 * (g4 & 1) + 4 * p) * 2;
        s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_PTR(s16x4))ad);
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if ((int)v[e] != (8 * h + 4 * half + e) * 32 + r) ++nbad;
    }
    if (nbad) atomicAdd(bad, nbad);
}
// Kernel-shaped variant: rows at an arbitrary pitch/base, row gather at stride (the wgrad G operand).
__global__ void selftest_tr16b_kernel(int* bad, int* info, int pitch, int base, int rowstride) {
    extern __shared__ __attribute__((aligned(16))) char sm[];
    const int lane = threadIdx.x;
    short* t = reinterpret_cast<short*>(sm);
    for (int i = lane; i < 8192; i += 64) t[i] = (short)(i * 7 + 3);
    __syncthreads();
    const int g4 = lane >> 4, li = lane & 15, q = li >> 2, p = li & 3, r = lane & 31, h = lane >> 5;
    int nbad = 0;
    for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            const int k = ks * 16 + 8 * (g4 >> 1) + 4 * half + q;
            const char* ad = sm + base + (k * rowstride) * pitch + (16 * (g4 & 1) + 4 * p) * 2;
            s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_PTR(s16x4))ad);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int kk = ks * 16 + 8 * h + 4 * half + e;
                const short want = *reinterpret_cast<const short*>(sm + base + (kk * rowstride) * pitch + r * 2);
                if (v[e] != want) { if (nbad == 0 && lane < 64) { info[lane * 4] = ks * 100 + half * 10 + e; info[lane * 4 + 1] = v[e]; info[lane * 4 + 2] = want; } ++nbad; }
            }
        }
    }
    if (nbad) atomicAdd(bad, nbad);
}
extern "C" int vae_selftest_tr16(vae_stream_t stream) {
    int* d = nullptr; int h[1 + 256];
    HIP_CHECK_RET(hipMalloc(&d, sizeof(h)));
    const int cfgs[5][3] = {{64, 0, 1}, {144, 1536, 1}, {80, 768, 1}, {144, 1536, 2}, {80, 768, 3}};
    std::string msg;
    HIP_CHECK_RET(hipMemsetAsync(d, 0, sizeof(h), (hipStream_t)stream));
    hipLaunchKernelGGL(selftest_tr16_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, d);
    HIP_CHECK_RET(hipMemcpyAsync(h, d, 4, hipMemcpyDeviceToHost, (hipStream_t)stream));
    HIP_CHECK_RET(hipStreamSynchronize((hipStream_t)stream));
    if (h[0]) msg += "basic:" + std::to_string(h[0]) + " ";
    for (int c = 0; c < 5; ++c) {
        HIP_CHECK_RET(hipMemsetAsync(d, 0, sizeof(h), (hipStream_t)stream));
        hipLaunchKernelGGL(selftest_tr16b_kernel, dim3(1), dim3(64), 16384, (hipStream_t)stream, d, d + 1, cfgs[c][0], cfgs[c][1], cfgs[c][2]);
        HIP_CHECK_RET(hipMemcpyAsync(h, d, sizeof(h), hipMemcpyDeviceToHost, (hipStream_t)stream));
        HIP_CHECK_RET(hipStreamSynchronize((hipStream_t)stream));
        if (h[0]) {
            msg += "cfg" + std::to_string(c) + ":" + std::to_string(h[0]) + "[";
            for (int l = 0; l < 64; l += 9) msg += "L" + std::to_string(l) + ":" + std::to_string(h[1 + l * 4]) + "," + std::to_string(h[2 + l * 4]) + "," + std::to_string(h[3 + l * 4]) + " ";
            msg += "] ";
        }
    }
    (void)hipFree(d);
    if (!msg.empty()) return vae_set_error("ds_read_b64_tr_b16 self-test", msg.c_str());
    return 0;
}
