// Context and the C ABI (include/vae_step.h) of the VAE step.  The launch sequencing is templated on the storage
// type and lives in vae_impl.cuh, instantiated by impl_bf16.hip / impl_f16.hip / impl_f32.hip.
#include <map>
#include <mutex>
#include "vae_ctx.h"
#include "edge_kernels.cuh"

static thread_local std::string g_err;
int vae_set_error(const char* what, const char* why) {
    g_err = std::string(what) + ": " + why;
    return -1;
}

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
    if (L < 1 || L > 4096) return vae_set_error("vae_param_layout", "latent_dim must be in 1..4096");
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

template <typename T> static T* dalloc(vae_ctx* c, size_t n) {
    void* p = nullptr;
    if (hipMalloc(&p, std::max<size_t>(n * sizeof(T), 256)) != hipSuccess) return nullptr;
    c->allocs.push_back(p); c->ws_bytes += (int64_t)std::max<size_t>(n * sizeof(T), 256);
    return reinterpret_cast<T*>(p);
}

// Side streams and the communication stream are per DEVICE, shared by every context of the process: each extra stream
// beyond HIP's hardware queues (GPU_MAX_HW_QUEUES) shares a queue with another and serialises with it - a second model
// in the process (evaluation copy, another batch size) used to slow the first one's step down by up to 3x.  Contexts of
// one device enqueue in host order, so sharing costs them nothing.  The streams live for the process.
struct DevStreams { hipStream_t side[vae_ctx::NSIDE]; hipStream_t comm; };
static DevStreams* device_streams(int side_prio) {
    static std::mutex mu;
    static std::map<int, DevStreams*> pool;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    std::lock_guard<std::mutex> lk(mu);
    auto it = pool.find(dev);
    if (it != pool.end()) return it->second;
    DevStreams* d = new DevStreams();
    bool ok = true;
    for (int i = 0; i < vae_ctx::NSIDE && ok; ++i) ok = hipStreamCreateWithPriority(&d->side[i], hipStreamNonBlocking, side_prio) == hipSuccess;
    {   // communication stream; VAE_COMM_STREAM_PRIO = high | low picks another priority class (diagnostics of the queue mapping)
        int lo = 0, hi = 0;
        (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
        const char* e = getenv("VAE_COMM_STREAM_PRIO");
        if (e && (!strcmp(e, "high") || !strcmp(e, "low"))) ok = ok && hipStreamCreateWithPriority(&d->comm, hipStreamNonBlocking, !strcmp(e, "high") ? hi : lo) == hipSuccess;
        else ok = ok && hipStreamCreateWithFlags(&d->comm, hipStreamNonBlocking) == hipSuccess;
    }
    if (!ok) { delete d; return nullptr; }
    pool[dev] = d;
    return d;
}

extern "C" void vae_destroy(vae_ctx* c) {
    if (!c) return;
    (void)vae_comm_destroy(c);
    for (void* p : c->allocs) (void)hipFree(p);
    if (c->n_side_ok) {
        // (the streams belong to the device pool; work of this context still on them is drained first)
        for (int i = 0; i < vae_ctx::NSIDE; ++i) { (void)hipStreamSynchronize(c->side[i]); (void)hipEventDestroy(c->ev_join[i]); }
        (void)hipStreamSynchronize(c->comm);
        for (int i = 0; i < vae_ctx::NFORK; ++i) (void)hipEventDestroy(c->ev_fork[i]);
        (void)hipEventDestroy(c->ev_pack); (void)hipEventDestroy(c->ev_comm);
        for (int i = 0; i < vae_ctx::NBUCKET; ++i) (void)hipEventDestroy(c->ev_bucket[i]);
    }
    delete c;
}
extern "C" int64_t vae_workspace_bytes(const vae_ctx* c) { return c ? c->ws_bytes : 0; }

extern "C" vae_ctx* vae_create(int H, int L, int maxB, int dtype, int gen) {
    vae_ctx* c = new vae_ctx();
    c->H = H; c->L = L; c->maxB = maxB; c->dtype = dtype; c->gen = gen; c->ws_bytes = 0; c->use_tr16 = 1; c->use_mfma_convout = 1; c->use_pipelined = 1; c->knob_up_per_cu = 2; c->knob_convout_grid = 1536; c->knob_convout_bwd_grid = 1024; c->knob_down_per_cu = 2; c->knob_nt_max = 4; c->knob_pipe_max_cout = 256; c->knob_ablate_b = 0; c->use_side_stream = 1; c->knob_bwd_per_cu = 0; c->knob_wave_nt_max = 4; c->knob_lay22_min_nt = 2; c->knob_down_waves = 8; c->knob_pack_grid = 128; c->knob_xcd_map = 1; c->knob_up_nt_max = 1; c->knob_lay42 = 1; c->knob_wgrad_layer_wgs = 0; c->knob_conv1_grid = 512; c->use_fused_bn = 1; c->knob_rev = 4; c->knob_lean = 1; c->walk_dir = 0; c->n_side_ok = 0; c->side_rr = 0; c->fork_rr = 0; c->comm_busy = 0; c->dbg_buf = nullptr; c->dbg_tag[0] = 0; c->dbg_epi = 0;
    if (getenv("VAE_NO_SIDE_STREAM")) c->use_side_stream = 0;   // diagnostics: everything on the caller's stream
    c->packed_for = nullptr; c->bwd_dirty = 1; c->bwd_half_done = 0; c->B = 0; c->trained = 0; c->prof = 0; c->tag = nullptr;
    if (vae_param_layout(H, L, gen, c->poff, c->psz, &c->ptotal) != 0) { delete c; return nullptr; }
    if (dtype != VAE_DTYPE_F32 && dtype != VAE_DTYPE_BF16 && dtype != VAE_DTYPE_F16) { vae_set_error("vae_create", "bad dtype"); delete c; return nullptr; }
    c->gmul = 1.f; c->ginv = 1.f; c->generic_accum = nullptr;
    if (maxB < 1) { vae_set_error("vae_create", "max_batch < 1"); delete c; return nullptr; }
    vae_bn_layout(c->bnoff, c->bnc, &c->bntotal);
    c->s = gen ? H / 16 : 2; c->s2 = c->s * c->s; c->F = 256LL * c->s2;
    c->npad_fc = (int)align_up(2 * L, 32); c->npad_di = (int)align_up(L, 32);
    c->esz = dtype == VAE_DTYPE_F32 ? 4 : 2;
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
        for (int wide = 0; wide < 2; ++wide)      // both tile shapes (the wide one exists for 16-bit storage only), small- and large-problem targets
            for (int big = 0; big < 2; ++big) {
                if (wide && dtype == VAE_DTYPE_F32) continue;
                for (int i = 1; i < 4; ++i) slab = std::max(slab, wgrad_slab_floats(c->wk, maxB, c->lay[i].H, c->lay[i].W, co[i], ci[i], &a, &b2, &wa, &wb, wide != 0, big != 0));
                for (int i = 4; i < 8; ++i) slab = std::max(slab, wgrad_slab_floats(c->wk, maxB, c->lay[i].H / 2, c->lay[i].W / 2, ci[i], co[i], &a, &b2, &wa, &wb, wide != 0, big != 0));
            }
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
            DevStreams* ds = device_streams(side_prio);
            ok = ds != nullptr;
            for (int i = 0; i < vae_ctx::NSIDE && ok; ++i) {
                c->side[i] = ds->side[i];
                ok = hipEventCreateWithFlags(&c->ev_join[i], hipEventDisableTiming) == hipSuccess;
            }
            for (int i = 0; i < vae_ctx::NFORK && ok; ++i) ok = hipEventCreateWithFlags(&c->ev_fork[i], hipEventDisableTiming) == hipSuccess;
            if (ok) { c->comm = ds->comm;
                      ok = hipEventCreateWithFlags(&c->ev_pack, hipEventDisableTiming) == hipSuccess &&
                           hipEventCreateWithFlags(&c->ev_comm, hipEventDisableTiming) == hipSuccess;
                      for (int i = 0; i < vae_ctx::NBUCKET && ok; ++i) ok = hipEventCreateWithFlags(&c->ev_bucket[i], hipEventDisableTiming) == hipSuccess; }
            c->n_side_ok = ok ? 1 : 0;
        }
    }
    if (ok && dtype != VAE_DTYPE_F32) {   // materialised operands of the deep weight gradients (small tensors)
        for (int i = 1; i <= 5 && ok; ++i) {
            const size_t n = B * c->lay[i].H * c->lay[i].W * c->lay[i].C;
            if (i == 1 || i == 2 || i == 4) { c->lay[i].act = dalloc<char>(c, n * c->esz); ok = ok && c->lay[i].act; }
            if (i >= 2) { c->lay[i].dy = dalloc<char>(c, n * c->esz); ok = ok && c->lay[i].dy; }
        }
    }
    if (ok) { c->reduce_tmp_floats = 64 * 1024; c->reduce_tmp = dalloc<float>(c, c->reduce_tmp_floats); ok = c->reduce_tmp != nullptr; }
    if (ok && dtype != VAE_DTYPE_F32) {
        c->fused_slab_floats = (size_t)512 * 9 * 64 * 32;   // up to 512 workgroups x [9][64][32]
        for (int i = 0; i < 3 && ok; ++i) { c->fused_slab[i] = dalloc<float>(c, c->fused_slab_floats); ok = c->fused_slab[i] != nullptr; }
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
    if (!strcmp(name, "knob_down_waves")) { c->knob_down_waves = value; return 0; }
    if (!strcmp(name, "knob_wgrad_layer_wgs")) { c->knob_wgrad_layer_wgs = value; return 0; }
    if (!strcmp(name, "knob_lay42")) { c->knob_lay42 = value; return 0; }
    if (!strcmp(name, "knob_up_nt_max")) { c->knob_up_nt_max = value < 1 ? 1 : value; return 0; }
    if (!strcmp(name, "knob_xcd_map")) { c->knob_xcd_map = value; return 0; }
    if (!strcmp(name, "knob_pack_grid")) { c->knob_pack_grid = value > 0 ? value : 1; return 0; }
    if (!strcmp(name, "knob_conv1_grid")) { c->knob_conv1_grid = value; return 0; }
    if (!strcmp(name, "knob_rev")) { c->knob_rev = value; return 0; }
    if (!strcmp(name, "knob_lean")) { c->knob_lean = value; return 0; }   // bit 0: noise beside conv1; 1: BN backward inside conv1_wgrad; 2: deferred loss on a side stream
    if (!strcmp(name, "use_fused_wgrad")) { c->use_fused_wgrad = value == 1 ? 3 : (value == 2 ? 1 : (value == 3 ? 2 : 0)); return 0; }   // 1 all, 2 decoder side only, 3 encoder.1 only
    if (!strcmp(name, "use_recomp_dz")) { c->use_recomp_dz = value; return 0; }
    if (!strcmp(name, "use_fc_dgrad8")) { c->use_fc_dgrad8 = value; return 0; }
    if (!strcmp(name, "use_fused_convout")) { c->use_fused_convout = value; return 0; }
    if (!strcmp(name, "use_convout_stream")) { c->use_convout_stream = value; return 0; }
    if (!strcmp(name, "use_wgrad_split")) { c->use_wgrad_split = value; return 0; }
    if (!strcmp(name, "use_upf_stream")) { c->use_upf_stream = value; return 0; }
    if (!strcmp(name, "use_dnf_stream")) { c->use_dnf_stream = value; return 0; }
    if (!strcmp(name, "knob_convout_bands")) { c->knob_convout_bands = value; return 0; }
    if (!strcmp(name, "knob_convout_step_grid")) { c->knob_convout_step_grid = value > 0 ? value : 1; return 0; }
    if (!strcmp(name, "knob_ablate_f")) { c->knob_ablate_f = value; return 0; }
    if (!strcmp(name, "knob_skip_wgrad")) { c->knob_skip_wgrad = value; return 0; }
    if (!strcmp(name, "use_raw_wgrad")) { c->use_raw_wgrad = value; return 0; }
    if (!strcmp(name, "use_deep")) { c->use_deep = value; return 0; }
    if (!strcmp(name, "use_latent_mfma")) { c->use_latent_mfma = value; return 0; }
    if (!strcmp(name, "knob_fused_grid")) { c->knob_fused_grid = std::max(1, std::min(value, 512)); return 0; }
    if (!strcmp(name, "knob_wgrad_tile")) { c->wk.tile = value; return 0; }
    if (!strcmp(name, "knob_wgrad_wide")) { c->wk.wide = value; return 0; }
    if (!strcmp(name, "knob_wgrad_mid8")) { c->wk.mid8 = value; return 0; }
    if (!strcmp(name, "knob_wgrad_force_simple")) { c->wk.force_simple = value; return 0; }   // diagnostics: the 64-bit-offset fallback kernel
    if (!strcmp(name, "knob_wgrad_wide_wgs")) { c->wk.wide_wgs = std::min(value, 1024); return 0; }
    if (!strcmp(name, "knob_wgrad_wgs")) { c->wk.wgs = std::min(value, 1024); return 0; }
    if (!strcmp(name, "knob_wgrad_cap_mb")) { c->wk.cap_mb = std::min(value, 48); return 0; }
    return vae_set_error("vae_set_option", "unknown option");
}
SideFork fork_side(vae_ctx* c, hipStream_t st, int which) {
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
int join_sides(vae_ctx* c, hipStream_t st) {
    if (!c->use_side_stream) return 0;
    for (int i = 0; i < vae_ctx::NSIDE; ++i) {
        HIP_CHECK_RET(hipEventRecord(c->ev_join[i], c->side[i]));
        HIP_CHECK_RET(hipStreamWaitEvent(st, c->ev_join[i], 0));
    }
    return 0;
}
// join the communication stream lent out by vae_comm_stream (work the caller enqueued on it, e.g. an all-reduce)
int join_comm(vae_ctx* c, hipStream_t st) {
    if (!c->comm_busy) return 0;
    HIP_CHECK_RET(hipEventRecord(c->ev_comm, c->comm));
    HIP_CHECK_RET(hipStreamWaitEvent(st, c->ev_comm, 0));
    c->comm_busy = 0;
    return 0;
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


// Byte / bit-plane stimuli -> the float32 batch the kernels read (train.py:630 copies float32 stimuli; pianorolls are 0/1 cells).
// kind 0: one byte per cell (uint8 / bool), value v -> (float)v; kind 1: bit planes, most significant bit first (numpy.packbits
// order), bit -> 0.f / 1.f.  The source may be device memory or PINNED host memory: the kernel then reads the batch straight over
// the host link (0.5 MB of bit planes or 4 MB of bytes for 256 images of 128x128), which removes the separate copy and its
// queue hand-over from the step's dependent chain.
__global__ void expand_bytes_kernel(const uint32_t* __restrict__ src, float* __restrict__ dst, long n4) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        const uint32_t w = __builtin_nontemporal_load(src + i);
        *reinterpret_cast<f32x4*>(dst + 4 * i) = f32x4{(float)(w & 255u), (float)((w >> 8) & 255u), (float)((w >> 16) & 255u), (float)(w >> 24)};
    }
}
__global__ void expand_bits_kernel(const uint8_t* __restrict__ src, float* __restrict__ dst, long nbytes) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < nbytes; i += (long)gridDim.x * blockDim.x) {
        const uint32_t w = __builtin_nontemporal_load(src + i);
        *reinterpret_cast<f32x4*>(dst + 8 * i) = f32x4{(float)((w >> 7) & 1u), (float)((w >> 6) & 1u), (float)((w >> 5) & 1u), (float)((w >> 4) & 1u)};
        *reinterpret_cast<f32x4*>(dst + 8 * i + 4) = f32x4{(float)((w >> 3) & 1u), (float)((w >> 2) & 1u), (float)((w >> 1) & 1u), (float)(w & 1u)};
    }
}
extern "C" int vae_expand_stimuli(const void* src, int kind, float* dst, int64_t n_cells, vae_stream_t stream) {
    if (!src || !dst) return vae_set_error("vae_expand_stimuli", "null pointer");
    if (kind != 0 && kind != 1) return vae_set_error("vae_expand_stimuli", "kind must be 0 (bytes) or 1 (bit planes)");
    if (n_cells < 0 || n_cells % 8) return vae_set_error("vae_expand_stimuli", "the number of cells must be a multiple of 8");
    if (n_cells == 0) return 0;
    // the address the device uses for the source: device memory as is, pinned host memory through its mapping; anything else
    // (pageable host memory) is refused - a kernel must not dereference it
    hipPointerAttribute_t at; memset(&at, 0, sizeof(at));
    if (hipPointerGetAttributes(&at, src) != hipSuccess) { (void)hipGetLastError(); return vae_set_error("vae_expand_stimuli", "the source is neither device nor pinned host memory"); }
    const void* dsrc = nullptr;
    if (at.type == hipMemoryTypeDevice || at.type == hipMemoryTypeManaged) dsrc = src;
    else if (at.type == hipMemoryTypeHost && at.devicePointer) dsrc = at.devicePointer;
    else return vae_set_error("vae_expand_stimuli", "the source is neither device nor pinned host memory");
    if ((reinterpret_cast<uintptr_t>(dsrc) & 3) || (reinterpret_cast<uintptr_t>(dst) & 15)) return vae_set_error("vae_expand_stimuli", "source must be 4-byte, destination 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    if (kind == 0) {
        const long n4 = n_cells / 4;
        hipLaunchKernelGGL(expand_bytes_kernel, dim3((unsigned)std::min<long>((n4 + 255) / 256, 2048)), dim3(256), 0, st, reinterpret_cast<const uint32_t*>(dsrc), dst, n4);
    } else {
        const long nb = n_cells / 8;
        hipLaunchKernelGGL(expand_bits_kernel, dim3((unsigned)std::min<long>((nb + 255) / 256, 2048)), dim3(256), 0, st, reinterpret_cast<const uint8_t*>(dsrc), dst, nb);
    }
    LAUNCH_CHECK("expand_stimuli_kernel");
    return 0;
}


// ---------------------------------------------------------------------------
extern "C" int vae_forward(vae_ctx* c, const float* x, int B, const float* params, float* bn_running, int64_t* nbt,
                           const float* eps, uint64_t seed, int train, float* xhat, float* mu, float* lv, float* z, vae_stream_t stream) {
    if (!c) return vae_set_error("vae_forward", "null ctx");
    if (B < 1 || B > c->maxB) return vae_set_error("vae_forward", "batch exceeds the context's max_batch");
    if (!x || !params || !xhat || !mu || !lv || !z) return vae_set_error("vae_forward", "null tensor pointer");
    hipStream_t st = (hipStream_t)stream;
    c->cur_stream = st; c->cur_stream_set = true;
    return VAE_DISPATCH(c->dtype, forward_impl, (c, x, B, params, bn_running, nbt, eps, seed, train, xhat, mu, lv, z, st));
}

extern "C" int vae_decode(vae_ctx* c, const float* z, int B, const float* params, float* bn_running, int64_t* nbt, int train,
                          float* xhat, vae_stream_t stream) {
    if (!c) return vae_set_error("vae_decode", "null ctx");
    if (B < 1 || B > c->maxB) return vae_set_error("vae_decode", "batch exceeds the context's max_batch");
    if (!z || !params || !xhat) return vae_set_error("vae_decode", "null tensor pointer");
    hipStream_t st = (hipStream_t)stream;
    c->B = B; c->trained = 0;   // a decode-only pass cannot be differentiated
    HIP_CHECK_RET(hipMemsetAsync(c->dstats, 0, c->n_dstats * sizeof(double), st)); c->bwd_dirty = 0;
    int rc = VAE_DISPATCH(c->dtype, pack_weights, (c, params, st));
    if (rc) return rc;
    // the reconstruction-loss side outputs of the output-conv kernel are unused here: xhat doubles as the target
    return VAE_DISPATCH(c->dtype, decode_impl, (c, z, B, params, bn_running, nbt, train, xhat, xhat, st));
}

extern "C" int vae_loss(vae_ctx* c, float kld_weight, float* out3, vae_stream_t stream) {
    if (!c || !c->B) return vae_set_error("vae_loss", "no forward");
    if (c->convout_pending) return vae_set_error("vae_loss", "the forward ran with train = 2: the ELBO is produced by the backward (use vae_loss_deferred)");
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
    if (c->convout_pending) { c->loss_out3 = out3; c->loss_kw = kld_weight; return 0; }   // finalised by the backward, after the fused output-conv kernel
    SideFork f = (c->knob_lean & 4) ? fork_side(c, (hipStream_t)stream) : SideFork{(hipStream_t)stream, c->slab, 0};
    if (f.rc) return f.rc;
    hipLaunchKernelGGL(loss_finalize_kernel, dim3(1), dim3(64), 0, f.st, c->accum, out3,
                       1.0 / ((double)c->B * c->H * c->H), 1.0 / (double)c->B, kld_weight, STAT_R);
    LAUNCH_CHECK("loss_finalize_kernel");
    return 0;
}

// Accumulators of vae_elbo_generic: the entry point has no context, so they live in a per-device ring (one slot per call:
// calls in flight on different streams of a device never share a slot unless more than kGenericSlots overlap).
static constexpr int kGenericSlots = 64, kMaxDevices = 64;
static double* g_generic_ring[kMaxDevices] = {nullptr};
static unsigned g_generic_next[kMaxDevices] = {0};
extern "C" int vae_elbo_generic(const float* xhat, const float* target, const float* mu, const float* lv, int64_t n, int B, int L,
                                float kld_weight, float* out3, float* g_xhat, float* g_mu, float* g_lv, vae_stream_t stream) {
    hipStream_t st = (hipStream_t)stream;
    int dev = 0;
    HIP_CHECK_RET(hipGetDevice(&dev));
    if (dev < 0 || dev >= kMaxDevices) return vae_set_error("vae_elbo_generic", "device index out of range");
    if (!g_generic_ring[dev]) HIP_CHECK_RET(hipMalloc(&g_generic_ring[dev], kGenericSlots * 4 * sizeof(double)));
    double* acc = g_generic_ring[dev] + 4 * (g_generic_next[dev]++ % kGenericSlots);
    HIP_CHECK_RET(hipMemsetAsync(acc, 0, 4 * sizeof(double), st));
    hipLaunchKernelGGL(bce_kernel, dim3((unsigned)std::min<long>((n + 255) / 256, 2048)), dim3(256), 0, st, xhat, target, g_xhat, acc, (long)n, (float)(1.0 / (double)n));
    LAUNCH_CHECK("bce_kernel");
    hipLaunchKernelGGL(kld_only_kernel, dim3((B * L + 255) / 256), dim3(256), 0, st, mu, lv, acc, B * L, kld_weight / (float)B, g_mu, g_lv);
    LAUNCH_CHECK("kld_only_kernel");
    hipLaunchKernelGGL(loss_finalize_kernel, dim3(1), dim3(64), 0, st, acc, out3, 1.0 / (double)n, 1.0 / (double)B, kld_weight, 1);
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
    c->cur_stream = st; c->cur_stream_set = true;
    return VAE_DISPATCH(c->dtype, backward_impl, (c, x, params, grads, g_xhat, gscale, g_mu, g_lv, g_z, g_pre, kld_weight, add_kl, part, st));
}
extern "C" int vae_backward(vae_ctx* c, const float* x, const float* params, float* grads, const float* g_xhat, const float* gscale,
                            const float* g_mu, const float* g_lv, const float* g_z, const float* g_pre, float kld_weight, int add_kl,
                            vae_stream_t stream) {
    return vae_backward_part(c, x, params, grads, g_xhat, gscale, g_mu, g_lv, g_z, g_pre, kld_weight, add_kl, 0, stream);
}

extern "C" int vae_adamw_step(float* params, const float* grads, float* m, float* v, int ngroups, const int64_t* offsets,
                              const int64_t* sizes, const double* lrs, const double* beta1s, double beta2, double eps, double weight_decay,
                              float grad_scale, int step, vae_stream_t stream) {
    if (ngroups < 1 || ngroups > 2) return vae_set_error("vae_adamw_step", "1 or 2 groups");
    if (step < 1) return vae_set_error("vae_adamw_step", "step is 1-based");
    AdamArgs a;
    // every scalar of torch's single-tensor update is formed in double from the Python floats and reaches the element-wise kernel as
    // one float: beta, 1 - beta, 1 - lr * weight_decay, lr / (1 - beta1^t), 1 / sqrt(1 - beta2^t), eps
    a.p = params; a.g = grads; a.m = m; a.v = v; a.ngrp = ngroups; a.beta2 = (float)beta2; a.omb2 = (float)(1.0 - beta2); a.eps = (float)eps;
    a.grad_scale = grad_scale; a.step = step;
    long nmax = 0;
    const double bc2 = 1.0 - pow(beta2, (double)step);
    for (int i = 0; i < ngroups; ++i) {
        a.grp[i].off = offsets[i]; a.grp[i].n = sizes[i]; nmax = std::max<long>(nmax, sizes[i]);
        a.grp[i].beta1 = (float)beta1s[i]; a.grp[i].omb1 = (float)(1.0 - beta1s[i]); a.grp[i].decay = (float)(1.0 - lrs[i] * weight_decay);
        const double bc1 = 1.0 - pow(beta1s[i], (double)step);   // torch.optim.AdamW: bias corrections with the CURRENT (cycled) beta1
        a.grp[i].step_size = (float)(lrs[i] / bc1); a.grp[i].inv_sqrt_bc2 = (float)(1.0 / sqrt(bc2));
    }
    hipLaunchKernelGGL(adamw_kernel, dim3((unsigned)std::min<long>((nmax / 4 + 255) / 256 + 1, 2048), ngroups), dim3(256), 0, (hipStream_t)stream, a);
    LAUNCH_CHECK("adamw_kernel");
    return 0;
}

extern "C" int vae_train_step(vae_ctx* c, const float* x, int B, float* params, float* grads, float* m, float* v, float* bn_running,
                              int64_t* nbt, const float* eps, uint64_t seed, float kld_weight, int ngroups, const int64_t* offsets,
                              const int64_t* sizes, const double* lrs, const double* beta1s, double beta2, double adam_eps,
                              double weight_decay, int step, float* xhat, float* mu, float* lv, float* z, float* out3, vae_stream_t stream) {
    if (vae_forward(c, x, B, params, bn_running, nbt, eps, seed, 1, xhat, mu, lv, z, stream)) return -1;
    if (vae_loss_deferred(c, kld_weight, out3, stream)) return -1;
    if (vae_backward(c, x, params, grads, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, kld_weight, 1, stream)) return -1;
    if (ngroups > 0 && vae_adamw_step(params, grads, m, v, ngroups, offsets, sizes, lrs, beta1s, beta2, adam_eps, weight_decay, 1.f, step, stream)) return -1;
    return 0;
}

// The fused training step as ONE host call (what torch_vae_amd.train.fused_step enqueues per iteration; train.py:634-659):
// forward with the output conv deferred (train = 2), ELBO scalars finalised beside the backward, backward, the gradient
// exchange of a data-parallel job, AdamW.  `exchange`:
//   0  none (single process);
//   1  ONE RCCL group over the `ngroups` optimised ranges on the compute stream, between the last backward kernel and AdamW;
//   2  bucketed: every group's all-reduce runs on the context's communication stream as soon as its gradients are complete
//      (the LAST group of the list - the decoder - after the first half of the backward, under the encoder half; the others
//      after the second half), and every group's AdamW launch waits only for its own bucket's event, so the update of one
//      group runs while the other group's all-reduce is still in flight (train.py:165-166,201,663 prepare this data-parallel
//      layout; the reference itself never exchanges).  Same arithmetic as 1: results are bit-identical.
// Modes 1 and 2 need vae_comm_init.  Outputs (xhat, mu, log_var, z, out3) are caller-owned as in vae_forward / vae_loss.
extern "C" int vae_train_step_fused(vae_ctx* c, const float* x, int B, float* params, float* grads, float* m, float* v, float* bn_running,
                                    int64_t* nbt, const float* eps, uint64_t seed, float kld_weight, int ngroups, const int64_t* offsets,
                                    const int64_t* sizes, const double* lrs, const double* beta1s, double beta2, double adam_eps,
                                    double weight_decay, float grad_scale, int step, int exchange, float* xhat, float* mu, float* lv,
                                    float* z, float* out3, vae_stream_t stream) {
    if (!c) return vae_set_error("vae_train_step_fused", "null ctx");
    if (exchange < 0 || exchange > 2) return vae_set_error("vae_train_step_fused", "exchange must be 0, 1 or 2");
    if (exchange && !c->nccl_comm) return vae_set_error("vae_train_step_fused", "gradient exchange without a communicator: call vae_comm_init first");
    if (exchange && ngroups < 1) return vae_set_error("vae_train_step_fused", "gradient exchange needs the optimised ranges");
    hipStream_t st = (hipStream_t)stream;
    if (vae_forward(c, x, B, params, bn_running, nbt, eps, seed, 2, xhat, mu, lv, z, stream)) return -1;
    if (vae_loss_deferred(c, kld_weight, out3, stream)) return -1;
    if (exchange != 2) {
        if (vae_backward(c, x, params, grads, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, kld_weight, 1, stream)) return -1;
        if (exchange == 1 && vae_allreduce_grads(c, grads, ngroups, offsets, sizes, 1, stream)) return -1;
        if (ngroups > 0 && vae_adamw_step(params, grads, m, v, ngroups, offsets, sizes, lrs, beta1s, beta2, adam_eps, weight_decay, grad_scale, step, stream)) return -1;
        return 0;
    }
    // bucketed exchange
    if (ngroups > vae_ctx::NBUCKET) return vae_set_error("vae_train_step_fused", "too many groups");
    if (vae_backward_part(c, x, params, grads, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, kld_weight, 1, 1, stream)) return -1;
    vae_stream_t cs = nullptr;
    const int last = ngroups - 1;
    if (vae_comm_stream(c, stream, &cs)) return -1;                              // ordered after the first half of the backward
    if (vae_allreduce_grads(c, grads, 1, offsets + last, sizes + last, 1, cs)) return -1;
    HIP_CHECK_RET(hipEventRecord(c->ev_bucket[last], (hipStream_t)cs));
    if (vae_backward_part(c, x, params, grads, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, kld_weight, 1, 2, stream)) return -1;   // (joins the communication stream)
    if (last > 0) {
        if (vae_comm_stream(c, stream, &cs)) return -1;                          // ordered after the second half
        if (vae_allreduce_grads(c, grads, last, offsets, sizes, 1, cs)) return -1;
        for (int i = 0; i < last; ++i) HIP_CHECK_RET(hipEventRecord(c->ev_bucket[i], (hipStream_t)cs));
    }
    for (int i = last; i >= 0; --i) {                                            // decoder first: its bucket has long arrived
        HIP_CHECK_RET(hipStreamWaitEvent(st, c->ev_bucket[i], 0));
        if (vae_adamw_step(params, grads, m, v, 1, offsets + i, sizes + i, lrs + i, beta1s + i, beta2, adam_eps, weight_decay, grad_scale, step, stream)) return -1;
    }
    c->comm_busy = 0;                                                            // every piece of work on the communication stream has been waited for
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
    struct Agg { std::string name; int calls; double ms, bytes, flops; int side; };
    std::vector<Agg> agg;
    for (auto& r : c->prof_recs) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, r.e0, r.e1) != hipSuccess) continue;
        Agg* a = nullptr;
        for (auto& x : agg) if (x.name == r.name) a = &x;
        if (!a) { agg.push_back({r.name, 0, 0, 0, 0, r.side}); a = &agg.back(); }
        a->calls += 1; a->ms += ms; a->bytes += r.bytes; a->flops += r.flops;
    }
    std::string out = "[";
    for (size_t i = 0; i < agg.size(); ++i) {
        char line[512];
        snprintf(line, sizeof(line), "%s{\"name\":\"%s\",\"calls\":%d,\"ms\":%.6f,\"bytes\":%.1f,\"flops\":%.1f,\"side\":%d}", i ? "," : "",
                 agg[i].name.c_str(), agg[i].calls, agg[i].ms, agg[i].bytes, agg[i].flops, agg[i].side);
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
    for (size_t i = 0; i < c->prof_recs.size(); ++i)
        for (int k = 0; k < c->prof_recs[i].launches; ++k) out += std::string(out.size() > 1 ? "," : "") + "\"" + c->prof_recs[i].name + "\"";
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
    return VAE_DISPATCH(c->dtype, pre_latents_impl, (c, out, (hipStream_t)stream));
}
extern "C" int vae_last_eps(vae_ctx* c, float* out, vae_stream_t stream) {
    if (!c || !c->B) return vae_set_error("vae_last_eps", "no forward");
    HIP_CHECK_RET(hipMemcpyAsync(out, c->eps, (size_t)c->B * c->L * 4, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return 0;
}

extern "C" int vae_debug_tensor(vae_ctx* c, int which, float* out, int64_t capacity, vae_stream_t stream) {
    if (!c || !c->B) return vae_set_error("vae_debug_tensor", "no forward");
    const void* src; int C, HW;
    if (which >= 0 && which < 16) { const BnLayer& l = c->lay[which & 7]; src = which < 8 ? l.y : l.dz; C = l.C; HW = l.H * l.W; }
    else if (which == 16 || which == 17) { src = which == 16 ? c->d0 : c->dd0; C = 256; HW = c->s2; }
    else return vae_set_error("vae_debug_tensor", "bad tensor id");
    const long n = (long)c->B * C * HW;
    if (n > capacity) return vae_set_error("vae_debug_tensor", "output too small");
    return VAE_DISPATCH(c->dtype, debug_tensor_impl, (c, src, out, n, C, HW, (hipStream_t)stream));
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
