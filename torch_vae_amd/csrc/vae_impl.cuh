// Launch sequencing of the VAE step, templated on the storage type T (bf16 / f16 / float).  Included by one
// translation unit per storage type, which instantiates the entry points declared at the end of vae_ctx.h.
#pragma once
#include "vae_ctx.h"
#include "conv_mfma.cuh"
#include "conv_pipe.cuh"
#include "edge_kernels.cuh"
#include "conv_fused.cuh"
#include "conv_deep.cuh"
#include "latent_mfma.cuh"
#include "convout_stream.cuh"
#include "wgrad_split.cuh"
#include "upfinal_stream.cuh"
#include "dnfirst_stream.cuh"

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


// Workgroup-specialised kernels of the deep layers (conv_deep.cuh): 16-bit storage, 128-pixel tiles of 8x16 pixels or two 8x8
// images, 128 (down) / 64 (up) output channels per workgroup.  Returns 1 when the launch is outside their domain (the caller
// then takes the pipelined kernels), 0 on success, -1 on error.
template <typename T>
static int launch_conv_deep(vae_ctx* c, ConvArgs<T> a, bool is_down, hipStream_t st) {
    if constexpr (sizeof(T) != 2) return 1;
    else {
        if (!(c->use_deep & (is_down ? 1 : 2)) || !fits_i32(a) || a.stage_out || c->knob_ablate_b) return 1;
        const int NCO = is_down ? 128 : 64;
        if (a.Cout % NCO || a.Cin % 32 || a.Cin > 256) return 1;
        if (((a.two_src & 1) != 0) != (a.epi != EPI_FWD)) return 1;
        if (!is_down && a.epi == EPI_PLAIN) return 1;
        Tiling t = make_tiling(a.Hs, a.Ws, 128);
        const int TB = 1 << t.lTB, th = 1 << t.lth, tw = 1 << t.ltw;
        if (!((tw == 16 && th == 8 && TB == 1) || (tw == 8 && th == 8 && TB == 2))) return 1;
        if ((a.two_src & 1) && a.slope != 1.f) return vae_set_error("conv_deep", "gradient operands are loaded without LeakyReLU (slope must be 1)");
        a.lth = t.lth; a.ltw = t.ltw; a.lTB = t.lTB; a.tiles_x = t.tiles_x; a.tiles_y = t.tiles_y;
        a.m_tx = fastdiv_magic(t.tiles_x); a.m_txy = fastdiv_magic(t.tiles_x * t.tiles_y);
        const int n_mt = ((a.B + TB - 1) / TB) * t.tiles_x * t.tiles_y, ntn = a.Cout / NCO, n_pairs = n_mt * ntn;
        a.n_mt = n_mt; a.rev = ((c->knob_rev >> 2) & 1) ? ((a.epi == EPI_FWD) ? ((c->knob_rev >> 4) & 1) : 1) : 0;
        DeepConvArgs<T> d; memset(&d, 0, sizeof(d));
        // LDS patch rows: a 32-pixel fragment read must touch 16 distinct 16-byte bank groups (conv_deep.cuh)
        const int PH = is_down ? 2 * th + 1 : th + 1;
        if (is_down) { d.rowp = tw == 16 ? 40 : 20; d.halfw = tw + 1; }
        else { d.rowp = tw == 16 ? 32 : 24; d.halfw = 0; }
        d.imgp = PH * d.rowp; d.npl = TB * d.imgp;
        d.m_rowp = fastdiv_magic(d.rowp); d.m_imgp = fastdiv_magic(d.imgp);
        if (d.npl > 64 * (is_down ? 12 : 7)) return vae_set_error("conv_deep", "patch larger than the producers' slot table");
        int grid = std::min(n_pairs, 256);
        grid = std::max(ntn, grid / ntn * ntn);   // a workgroup keeps one N tile (register-resident statistics)
        a.xcd = (c->knob_xcd_map && grid % 8 == 0 && (grid / 8) % ntn == 0) ? grid / 8 : 0;
        const size_t lds = (size_t)((3 * a.Cin * 4 + 15) & ~15) + 2 * (size_t)d.npl * 80 +
                           (is_down ? 2 * (size_t)(3 * 4 * 128 * 16) + 4 * 2 * 32 * 2 * 4 : 2 * (size_t)(9 * 4 * 64 * 16) + 4 * 32 * 2 * 4);
        if (is_down && (size_t)d.npl * 80 < 4 * 64 * 144) return vae_set_error("conv_deep", "patch half smaller than the epilogue tiles");
        if (lds > 160 * 1024) return vae_set_error("conv_deep", "tile does not fit LDS");
        a.dbg = (c->dbg_buf && is_down == !(c->dbg_epi & 16) && c->tag && !strcmp(c->tag, c->dbg_tag) && a.epi == (c->dbg_epi & 15)) ? c->dbg_buf : nullptr;
        d.c = a; d.ablate = c->knob_ablate_f;
        const double px_lo = (double)a.B * a.Hs * a.Ws, px_hi = 4 * px_lo;
        const double px_in = is_down ? px_hi : px_lo, px_out = is_down ? px_lo : px_hi;
        ProfScope ps(c, is_down ? (a.epi == EPI_FWD ? "down_fwd(conv)" : "down_bwd(convT dgrad)") : (a.epi == EPI_FWD ? "up_fwd(convT)" : "up_bwd(conv dgrad)"),
                     sizeof(T) * (px_in * a.Cin * ((a.two_src & 1) ? 2 : 1) + px_out * a.Cout * (a.epi == EPI_BWD ? 2 : 1) + 9.0 * a.Cin * a.Cout),
                     2.0 * 9 * a.Cin * a.Cout * px_lo, st);
#define DEEP_CASE(K, E) { if (set_lds(K<T, E>, lds)) return -1; hipLaunchKernelGGL((K<T, E>), dim3(grid), dim3(512), lds, st, d, n_pairs, ntn); }
        if (is_down) { if (a.epi == EPI_FWD) DEEP_CASE(dn3_kernel, EPI_FWD) else if (a.epi == EPI_BWD) DEEP_CASE(dn3_kernel, EPI_BWD) else DEEP_CASE(dn3_kernel, EPI_PLAIN) }
        else { if (a.epi == EPI_FWD) DEEP_CASE(up3_kernel, EPI_FWD) else DEEP_CASE(up3_kernel, EPI_BWD) }
#undef DEEP_CASE
        LAUNCH_CHECK("conv_deep_kernel");
        return 0;
    }
}

template <typename T>
static int launch_down(vae_ctx* c, ConvArgs<T> a, hipStream_t st) {
    if constexpr (sizeof(T) == 2) {
        // encoder.1's forward on 128x128 images: the row-streaming kernel (dnfirst_stream.cuh)
        if (c->use_dnf_stream && a.epi == EPI_FWD && a.Cin == 32 && a.Cout == 64 && a.Hs == dfs::HO && a.Ws == dfs::WO && !a.stage_out && !a.two_src) {
            DnFirstStreamArgs<T> m;
            m.yin = a.src0; m.coef = a.coef; m.slope = a.slope; m.fuse = a.fuse; m.wp = a.wp; m.bias = a.bias; m.out = a.out; m.stat = a.stat; m.B = a.B;
            const int ncu = 256;
            long best = -1; int nb = 1;
            for (int cand = 1; cand <= 8 && dfs::HO / cand >= 4; cand *= 2) {
                const long rounds = ((long)a.B * cand + ncu - 1) / ncu, cost = rounds * (dfs::HO / cand / 2 + 2);
                if (best < 0 || cost < best) { best = cost; nb = cand; }
            }
            m.nb = nb; m.RB = dfs::HO / nb; m.n_units = a.B * nb;
            const double px_out = (double)a.B * a.Hs * a.Ws;
            ProfScope ps(c, "down_fwd(conv)", sizeof(T) * (4 * px_out * 32 + px_out * 64 + 9.0 * 32 * 64), 2.0 * 9 * 32 * 64 * px_out, st);
            const size_t lds = dnfirst_stream_lds();
            if (set_lds(dnfirst_stream_kernel<T>, lds)) return -1;
            hipLaunchKernelGGL((dnfirst_stream_kernel<T>), dim3(std::min(m.n_units, ncu)), dim3(768), lds, st, m);
            LAUNCH_CHECK("dnfirst_stream_kernel");
            return 0;
        }
    }
    if (c->use_pipelined) { const int rc = launch_conv_deep<T>(c, a, true, st); if (rc <= 0) return rc; }
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
    if constexpr (sizeof(T) == 2) {
        // final_layer.0's forward on 128x128 images: the row-streaming kernel (upfinal_stream.cuh)
        const bool upf7 = a.Cin == 32 && a.Hs == 64 && a.Ws == 64, upf6 = a.Cin == 64 && a.Hs == 32 && a.Ws == 32;   // final_layer.0 / decoder.2 at 128x128
        // (decoder.2 - bit 1 of the option - measures the same 32 us as the tiled kernel: off by default)
        if (a.epi == EPI_FWD && a.Cout == 32 && ((upf7 && (c->use_upf_stream & 1)) || (upf6 && (c->use_upf_stream & 2))) && !a.stage_out && !a.two_src) {
            const int HLr = a.Hs;
            UpFinalStreamArgs<T> m;
            m.yin = a.src0; m.coef = a.coef; m.slope = a.slope; m.fuse = a.fuse; m.wp = a.wp; m.bias = a.bias; m.out = a.out; m.stat = a.stat; m.B = a.B;
            const int ncu = 256;
            long best = -1; int nb = 1;
            for (int cand = 1; cand <= 8 && HLr / cand >= 8; cand *= 2) {
                const long rounds = ((long)a.B * cand + ncu - 1) / ncu, cost = rounds * (HLr / cand / 4 + 2);
                if (best < 0 || cost < best) { best = cost; nb = cand; }
            }
            m.nb = nb; m.RB = HLr / nb; m.n_units = a.B * nb;
            const double px_in = (double)a.B * a.Hs * a.Ws;
            ProfScope ps(c, "up_fwd(convT)", sizeof(T) * (px_in * a.Cin + 4 * px_in * 32 + 9.0 * a.Cin * 32), 2.0 * 9 * a.Cin * 32 * px_in, st);
            if (upf7) {
                const size_t lds = upfinal_stream_lds<32, 64>();
                if (set_lds(upfinal_stream_kernel<T, 32, 64>, lds)) return -1;
                hipLaunchKernelGGL((upfinal_stream_kernel<T, 32, 64>), dim3(std::min(m.n_units, ncu)), dim3(1024), lds, st, m);
            } else {
                const size_t lds = upfinal_stream_lds<64, 32>();
                if (set_lds(upfinal_stream_kernel<T, 64, 32>, lds)) return -1;
                hipLaunchKernelGGL((upfinal_stream_kernel<T, 64, 32>), dim3(std::min(m.n_units, ncu)), dim3(1024), lds, st, m);
            }
            LAUNCH_CHECK("upfinal_stream_kernel");
            return 0;
        }
    }
    if (c->use_pipelined) { const int rc = launch_conv_deep<T>(c, a, false, st); if (rc <= 0) return rc; }
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
    if (!is_down) NT = std::min(NT, c->knob_up_nt_max);
    if (!is_down && (a.epi == EPI_BWD || sizeof(T) == 4)) NT = 1;
    // tile organisation: 2x2 wave grid over a 128-pixel workgroup tile (wide down tiles: halves the weight-fragment
    // traffic), wave-independent 32-pixel tiles (no workgroup barrier in the loop), or one row of waves per workgroup tile
    // (16-bit storage only: f32 keeps the row-of-waves tile.  NT = 2 - encoder.1's forward - measured 47 vs 52 us against the
    //  wave-independent tiles)
    const bool lay22 = is_down && sizeof(T) == 2 && NT >= 2 && NT >= c->knob_lay22_min_nt;
    const bool lay24 = lay22 && NT == 4 && sizeof(T) == 2 && c->knob_down_waves == 8;    // eight waves: 2 x 4 grid
    const bool lay42 = lay22 && NT == 2 && sizeof(T) == 2 && c->knob_down_waves == 8 && (c->knob_lay42 != 0);   // eight waves: 4 x 2 grid
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
    const size_t lds = ((3 * a.Cin * 4 + 15) & ~15) + (size_t)(wv ? 4 : 1) * TB * PHW * PATCH_PITCH + (lay24 ? 512 * (8 * NT * sizeof(T) + 16) : lay42 ? 256 * (16 * NT * sizeof(T) + 16) : lay22 ? 256 * (16 * NT * sizeof(T) + 16) : (is_down ? 128 : 256) * opitch) + 4 * NT * 32 * 2 * 4 +
                       std::max<size_t>((size_t)TB * PHW * 4, (lay24 || lay42) ? (size_t)5 * 512 : (size_t)(is_down ? 10 : 3) * (wv ? 64 : 256)) * 8;   // + the per-item staging table (padded to MAXI*SSTR)
    if (lds > 160 * 1024) return vae_set_error("conv_pipe", "tile does not fit LDS");
    if (c->knob_ablate_b) a.two_src |= 2;
    a.dbg = (c->dbg_buf && is_down == !(c->dbg_epi & 16) && c->tag && !strcmp(c->tag, c->dbg_tag) && a.epi == (c->dbg_epi & 15)) ? c->dbg_buf : nullptr;
    if ((a.two_src & 1) && a.slope != 1.f) return vae_set_error("conv_pipe", "gradient operands are loaded without LeakyReLU (slope must be 1)");
    const int per_cu = (int)std::max<size_t>(1, std::min<size_t>(is_down ? c->knob_down_per_cu : c->knob_up_per_cu, (160 * 1024) / lds));
    const int n_wg_pairs = wv ? ((n_mt + 3) / 4) * ntn : n_pairs;    // workgroup-level work items
    int grid = std::min(n_wg_pairs, 256 * ((c->knob_bwd_per_cu > 0 && a.epi != EPI_FWD) ? std::min(per_cu, c->knob_bwd_per_cu) : per_cu));
    grid = std::max(ntn, grid / ntn * ntn);   // a workgroup must stay on one N tile (register-resident statistics)
    a.xcd = (c->knob_xcd_map && grid % 8 == 0 && (grid / 8) % ntn == 0) ? grid / 8 : 0;   // contiguous id range per XCD (conv_pipe.cuh: vb)
    const double px_lo = (double)a.B * a.Hs * a.Ws, px_hi = 4 * px_lo;
    const double px_in = is_down ? px_hi : px_lo, px_out = is_down ? px_lo : px_hi;
    ProfScope ps(c, is_down ? (a.epi == EPI_FWD ? "down_fwd(conv)" : "down_bwd(convT dgrad)") : (a.epi == EPI_FWD ? "up_fwd(convT)" : "up_bwd(conv dgrad)"),
                 sizeof(T) * (px_in * a.Cin * ((a.two_src & 1) ? 2 : 1) + px_out * a.Cout * (a.epi == EPI_BWD ? 2 : 1) + 9.0 * a.Cin * a.Cout),
                 2.0 * 9 * a.Cin * a.Cout * px_lo, st);
    if (((a.two_src & 1) != 0) != (a.epi != EPI_FWD)) return vae_set_error("conv_pipe", "forward launches stage one source, backward launches two");
#define PIPE_CASE(K, N, E, V) { if (set_lds(K<T, N, E, V>, lds)) return -1; hipLaunchKernelGGL((K<T, N, E, V>), dim3(grid), dim3(256), lds, st, a, n_pairs, ntn); }
#define PIPE_CASE22(N, E) { if (set_lds(down2_kernel<T, N, E, false, 1>, lds)) return -1; hipLaunchKernelGGL((down2_kernel<T, N, E, false, 1>), dim3(grid), dim3(256), lds, st, a, n_pairs, ntn); }
#define PIPE_CASE42(E) { if (set_lds(down2_kernel<T, 2, E, false, 3>, lds)) return -1; hipLaunchKernelGGL((down2_kernel<T, 2, E, false, 3>), dim3(grid), dim3(512), lds, st, a, n_pairs, ntn); }
#define PIPE_CASE24(E) { if (set_lds(down2_kernel<T, 4, E, false, 2>, lds)) return -1; hipLaunchKernelGGL((down2_kernel<T, 4, E, false, 2>), dim3(grid), dim3(512), lds, st, a, n_pairs, ntn); }
#define PIPE_WV(K, N, E) { if constexpr (sizeof(T) == 2) { if (wv) PIPE_CASE(K, N, E, true) else PIPE_CASE(K, N, E, false) } else PIPE_CASE(K, N, E, false) }
#define PIPE_EPI(K, N) { if (a.epi == EPI_FWD) PIPE_WV(K, N, EPI_FWD) else if (a.epi == EPI_BWD) PIPE_WV(K, N, EPI_BWD) else PIPE_WV(K, N, EPI_PLAIN) }
#define PIPE_EPI22(N) { if (a.epi == EPI_FWD) PIPE_CASE22(N, EPI_FWD) else if (a.epi == EPI_BWD) PIPE_CASE22(N, EPI_BWD) else PIPE_CASE22(N, EPI_PLAIN) }
    if (lay24) { if constexpr (sizeof(T) == 2) { if (a.epi == EPI_FWD) PIPE_CASE24(EPI_FWD) else if (a.epi == EPI_BWD) PIPE_CASE24(EPI_BWD) else PIPE_CASE24(EPI_PLAIN) } }
    else if (lay42) { if constexpr (sizeof(T) == 2) { if (a.epi == EPI_FWD) PIPE_CASE42(EPI_FWD) else if (a.epi == EPI_BWD) PIPE_CASE42(EPI_BWD) else PIPE_CASE42(EPI_PLAIN) } }
    else if (lay22) { if (NT == 2) PIPE_EPI22(2) else { if constexpr (sizeof(T) == 2) PIPE_EPI22(4) } }
    else if (is_down) { if (NT == 1) PIPE_EPI(down2_kernel, 1) else if (NT == 2) PIPE_EPI(down2_kernel, 2) else PIPE_EPI(down2_kernel, 4) }
    else if (a.epi == EPI_FWD) { if (NT == 1) PIPE_WV(up2_kernel, 1, EPI_FWD) else PIPE_WV(up2_kernel, 2, EPI_FWD) }
    else if (a.epi == EPI_BWD) PIPE_WV(up2_kernel, 1, EPI_BWD)
    else return vae_set_error("conv_pipe", "up kernel has no plain epilogue");
#undef PIPE_EPI22
#undef PIPE_EPI
#undef PIPE_WV
#undef PIPE_CASE42
#undef PIPE_CASE24
#undef PIPE_CASE22
#undef PIPE_CASE
    LAUNCH_CHECK("conv_pipe_kernel");
    return 0;
}

// (every caller reduces a parameter gradient: the result is written times c->ginv, the inverse of the f16 gradient scale)
static int launch_reduce(const float* slab, int nslab, size_t n, float* out, int CA, int CB, hipStream_t st, vae_ctx* c) {
    ProfScope ps(c, "reduce_slab", 4.0 * n * (nslab + 1), 0, st);
    // many slabs of a small tensor (the output conv's 288 weights from 1536 workgroups): a handful of workgroups summing
    // them serially took 70-80 us; two levels: G partial sums per output, then the G partials
    if (nslab >= 256 && n * 32 * 4 <= c->reduce_tmp_floats) {
        const int G = 32, per = (nslab + G - 1) / G;
        // (reductions on different streams may be in flight together: each takes the next of the buffer's slots)
        constexpr size_t kSlot = 16384;     // fixed slot stride: reductions of different sizes must never overlap
        if (n * G > kSlot) return vae_set_error("reduce", "two-level scratch slot too small");
        float* tmp = c->reduce_tmp + (size_t)(c->reduce_slot++ % (c->reduce_tmp_floats / kSlot)) * kSlot;
        if (ps.idx >= 0) c->prof_recs[ps.idx].launches = 2;   // (vae_profile_sequence lists one entry per device launch)
        hipLaunchKernelGGL(reduce_slab_kernel, dim3((unsigned)((n + 63) / 64), G), dim3(256), 0, st, slab, nslab, (int)n, tmp, 0, 0, 1.f, per);
        LAUNCH_CHECK("reduce_slab_kernel");
        hipLaunchKernelGGL(reduce_slab_kernel, dim3((unsigned)((n + 63) / 64)), dim3(256), 0, st, tmp, G, (int)n, out, CA, CB, c->ginv, G);
        LAUNCH_CHECK("reduce_slab_kernel");
        return 0;
    }
    hipLaunchKernelGGL(reduce_slab_kernel, dim3((unsigned)((n + 63) / 64)), dim3(256), 0, st, slab, nslab, (int)n, out, CA, CB, c->ginv, nslab);
    LAUNCH_CHECK("reduce_slab_kernel");
    return 0;
}

// raw: both operands are materialised tensors (a.s0 low-res side, a.g0 high-res side), staged as plain copies
template <typename T>
static int launch_wgrad(vae_ctx* c, WgradArgs<T> a, float* dw_out, hipStream_t st, float* slab_buf = nullptr, bool raw = false) {
    if (!slab_buf) slab_buf = c->slab;
    int nsplit, tps, WA, WB;
    const bool big = (double)c->B * c->H * c->H >= (double)(1 << 21);   // e.g. 128x128 at batch >= 128
    // the prefetching variants index both operands with 32-bit BYTE offsets: tensors of 4 GiB or more (or the
    // knob_wgrad_force_simple diagnostic) take the synchronous kernel, which uses 64-bit element offsets
    const bool fits32 = 4.0 * a.B * a.Hs * a.Ws * std::max(a.CA, a.CB) * sizeof(T) < 4294967296.0 && !c->wk.force_simple;
    const bool pre = c->use_pipelined && sizeof(T) == 2 && fits32;
    WgradKnobs wk = c->wk;
    // per-layer override of the wide-tile split (diagnostic: knob_wgrad_layer_wgs = layer_mask << 16 | workgroups;
    // bit i of the mask = BN layer i, named by the current tag)
    if (c->knob_wgrad_layer_wgs && c->tag) {
        static const char* kTag[8] = {"encoder.0", "encoder.1", "encoder.2", "encoder.3", "decoder.0", "decoder.1", "decoder.2", "final_layer.0"};
        for (int i = 0; i < 8; ++i)
            if (!strcmp(c->tag, kTag[i]) && ((c->knob_wgrad_layer_wgs >> (16 + i)) & 1)) wk.wide_wgs = wk.wgs = c->knob_wgrad_layer_wgs & 0xffff;
    }
    const size_t need = wgrad_slab_floats(wk, a.B, a.Hs, a.Ws, a.CA, a.CB, &nsplit, &tps, &WA, &WB, pre, big);
    if (need > c->slab_floats) return vae_set_error("wgrad", "slab too small");
    Tiling t = make_tiling(a.Hs, a.Ws, WG_KP);
    a.lth = t.lth; a.ltw = t.ltw; a.lTB = t.lTB; a.tiles_x = t.tiles_x; a.tiles_y = t.tiles_y;
    const int TB = 1 << t.lTB, th = 1 << t.lth, tw = 1 << t.ltw;
    a.n_tiles = ((a.B + TB - 1) / TB) * t.tiles_x * t.tiles_y; a.tiles_per_split = tps;
    a.slab = slab_buf; a.use_tr16 = c->use_tr16; a.rev = (c->knob_rev >> 3) & 1;
    a.m_pp = fastdiv_magic((2 * th + 1) * (2 * tw + 1)); a.m_pw = fastdiv_magic(2 * tw + 1); a.m_tx = fastdiv_magic(t.tiles_x); a.m_txy = fastdiv_magic(t.tiles_x * t.tiles_y);
    const bool mid8 = c->wk.mid8 && WA == 2 && WB == 1 && pre;   // eight waves on the 64x32-channel tile
    const int nthr = (WA == 4 || mid8) ? 512 : 256, maxg = (5 * WB * 256 + nthr - 1) / nthr;
    const size_t lds = (size_t)(3 * 32 * WA + 3 * 32 * WB) * 4 + (size_t)WG_KP * (32 * WA * sizeof(T) + WG_SPAD) +
                       (size_t)TB * (2 * th + 1) * (2 * tw + 1) * (32 * WB * sizeof(T) + WG_GPAD) +
                       (pre ? std::max<size_t>((size_t)TB * (2 * th + 1) * (2 * tw + 1) * (32 * WB * sizeof(T) / 16), (size_t)maxg * nthr) * 8 : 0);   // + staging table (prefetching variants, padded to MAXG*threads)
    dim3 grid(nsplit, a.CA / (32 * WA), a.CB / (32 * WB));
    const double px_s = (double)a.B * a.Hs * a.Ws;
    {
    ProfScope ps(c, "wgrad_kernel",
                 sizeof(T) * (px_s * a.CA * (a.s_two ? 2 : 1) + 4 * px_s * a.CB * (a.g_two ? 2 : 1)) + 4.0 * 9 * a.CA * a.CB,
                 2.0 * 9 * a.CA * a.CB * px_s, st);
    // s_two/g_two identify the layer kind: Conv2d (gradient on the low-res side) or ConvTranspose2d
    if (raw && !(WA == 4 && pre)) return vae_set_error("wgrad", "materialised operands: wide prefetching tile only");
    if (!raw && a.s_two == a.g_two) return vae_set_error("wgrad", "exactly one operand must be the gradient");
    const bool convt = a.g_two != 0;
#define WG_CASE(A_, B_, C_, P_) { if (set_lds(wgrad_kernel<T, A_, B_, C_, P_>, lds)) return -1; hipLaunchKernelGGL((wgrad_kernel<T, A_, B_, C_, P_>), grid, dim3(256), lds, st, a); }
#define WG_KIND(A_, B_, P_) { if (convt) WG_CASE(A_, B_, true, P_) else WG_CASE(A_, B_, false, P_) }
    bool split_done = false;
    if constexpr (sizeof(T) == 2) {
        // producer / consumer form of the wide tile (wgrad_split.cuh): same results, staging and MFMA halves in different waves
        const int npix = TB * (2 * th + 1) * (2 * tw + 1);
        if (WA == 4 && pre && !raw && c->use_wgrad_split && c->use_tr16 && npix * (int)(32 * sizeof(T) / 16) <= wsp::MAXG * wsp::NP &&
            wgrad_split_lds<T>(npix) <= 160 * 1024) {
            a.dbg = (c->dbg_buf && c->tag && !strcmp(c->tag, c->dbg_tag) && c->dbg_epi == 32) ? c->dbg_buf : nullptr;   // (vae_debug_stamps(tag, 32, buf))
            const size_t lds2 = wgrad_split_lds<T>(npix);
            if (convt) { if (set_lds(wgrad_split_kernel<T, true>, lds2)) return -1; hipLaunchKernelGGL((wgrad_split_kernel<T, true>), grid, dim3(1024), lds2, st, a); }
            else { if (set_lds(wgrad_split_kernel<T, false>, lds2)) return -1; hipLaunchKernelGGL((wgrad_split_kernel<T, false>), grid, dim3(1024), lds2, st, a); }
            split_done = true;
        }
    }
    if (split_done) {}
    else if (WA == 4) {
        if constexpr (sizeof(T) == 2) {
            if (raw) { if (set_lds(wgrad_kernel<T, 4, 1, false, true, 8, true>, lds)) return -1; hipLaunchKernelGGL((wgrad_kernel<T, 4, 1, false, true, 8, true>), grid, dim3(512), lds, st, a); }
            else if (convt) { if (set_lds(wgrad_kernel<T, 4, 1, true, true, 8>, lds)) return -1; hipLaunchKernelGGL((wgrad_kernel<T, 4, 1, true, true, 8>), grid, dim3(512), lds, st, a); }
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
    // 32-column blocks per workgroup: the largest of {4, 2, 1} that divides Npad/32 (the kernel has no partial N tile)
    const int nb = a.Npad / 32, NT = nb % 4 == 0 ? 4 : (nb % 2 == 0 ? 2 : 1);
    if (a.Npad % 32) return vae_set_error("dense", "Npad must be a multiple of 32");
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
template <typename T>
int pack_weights(vae_ctx* c, const float* params, hipStream_t st) {
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
    hipLaunchKernelGGL((pack_kernel<T>), dim3((unsigned)c->knob_pack_grid, (unsigned)d.size()), dim3(256), 0, st, c->d_descs);
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
    f.C = l.C; f.count = (double)c->B * l.H * l.W; f.inv_count = 1.0 / f.count; f.eps = kBnEps; f.momentum = kBnMom; f.update_running = bn_running != nullptr;
    f.mode = BNF_FWD;
    return f;
}
static BnFuse make_fuse_bwd(vae_ctx* c, int i, const float* params, float* grads) {
    const BnLayer& l = c->lay[i];
    BnFuse f; memset(&f, 0, sizeof(f));
    f.stat = l.stat_b; f.gamma = params + c->poff[l.p_gamma]; f.block = l.block;
    f.dgamma = grads + c->poff[l.p_gamma]; f.dbeta = grads + c->poff[l.p_beta]; f.dconv_bias = grads + c->poff[l.p_convb];
    f.C = l.C; f.count = (double)c->B * l.H * l.W; f.inv_count = 1.0 / f.count; f.mode = BNF_BWD; f.ginv = c->ginv;
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
template <typename T>
int decode_impl(vae_ctx* c, const float* z, int B, const float* params, float* bn_running, int64_t* nbt, int train,
                       const float* x, float* xhat, hipStream_t st) {
    const int H = c->H, L = c->L;
    static const char* kLayerTag[8] = {"encoder.0", "encoder.1", "encoder.2", "encoder.3", "decoder.0", "decoder.1", "decoder.2", "final_layer.0"};
    c->tag = "latent";
    // decoder_input
    if (c->use_latent_mfma & 1) {   // 64 feature columns x the whole batch per workgroup on the exact-f32 MFMA (latent_mfma.cuh)
        RowGemmArgs g; memset(&g, 0, sizeof(g));
        g.Y = z; g.ldy = L; g.K = L; g.Wf = params + c->poff[20]; g.bias = params + c->poff[21]; g.out = c->d0; g.B = B; g.F = (int)c->F; g.s2 = c->s2;
        ProfScope ps(c, "decin_fwd", (double)sizeof(T) * B * (double)c->F + 4.0 * c->F * L, 2.0 * B * c->F * L, st);
        const size_t lds = row_gemm_lds<T>();
        if (set_lds(row_gemm_kernel<T, 0>, lds)) return -1;
        hipLaunchKernelGGL((row_gemm_kernel<T, 0>), dim3((unsigned)(c->F / 64)), dim3(256), lds, st, g);
        LAUNCH_CHECK("row_gemm_kernel");
    } else {
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
        if (i == 5) {
            c->lay[4].act_ok = 0;
            if (train && c->lay[4].act && raw_wgrad_ok<T>(c, 5) && will_pipe(c, a)) { a.stage_out = reinterpret_cast<T*>(c->lay[4].act); c->lay[4].act_ok = 1; }
        }
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
        c->convout_pending = 0; c->loss_out3 = nullptr; c->dlogit_valid = 1;
        if (train == 2 && mfma_out && c->use_fused_convout && c->use_fused_bn && !c->use_recomp_dz && f7.mode == BNF_FWD && H % 32 == 0) {
            // fused training step: forward AND backward of this layer run as one kernel at the start of the backward
            c->pending_f7 = f7; c->convout_pending = 1; c->dlogit_valid = 0;
            return 0;
        }
        ProfScope ps(c, "convout_fwd+bce", ((double)sizeof(T) * 32 + 12.0) * B * H * H, 2.0 * 9 * 32 * B * H * H, st);
        bool launched = false;
        if constexpr (sizeof(T) == 2) {
            if (mfma_out) {
                ConvOutFwdMfmaArgs<T> m; m.fuse = f7; m.rev = c->knob_rev & 1;
                m.yf = reinterpret_cast<const T*>(c->lay[7].y); m.coef = a.coef; m.wt = a.wt; m.bias = a.bias; m.target = x;
                m.xhat = xhat; m.dlogit = c->dlogit; m.accum = c->accum; m.B = B; m.H = H; m.W = H; m.n_tiles = B * (H / 8) * (H / 32);
                m.inv_n = a.inv_n; m.slope = kSlope;
                hipLaunchKernelGGL((convout_fwd_mfma_kernel<T>), dim3(std::min(m.n_tiles, c->knob_convout_grid)), dim3(256), 0, st, m);
                launched = true;
            }
        }
        if (!launched) hipLaunchKernelGGL((convout_fwd_kernel<T>), dim3(B * (H / 16) * (H / 32)), dim3(256), 0, st, a);
        LAUNCH_CHECK("convout_fwd_kernel");
    }
    return 0;
}

template <typename T>
int forward_impl(vae_ctx* c, const float* x, int B, const float* params, float* bn_running, int64_t* nbt,
                        const float* eps, uint64_t seed, int train, float* xhat, float* mu, float* lv, float* z, hipStream_t st) {
    const int H = c->H, L = c->L;
    c->B = B; c->trained = train; c->x = x; c->xhat = xhat; c->mu = mu; c->lv = lv; c->z = z;
    // f16 storage: gradient scale for the backward of this forward (vae_ctx.h): dL/dlogit is O(1/(B*H*W)), far below the
    // smallest f16 normal; 2^ceil(log2(B*H*W)) / 16 puts the stored dz around 2^-4, mid-range
    c->gmul = 1.f; c->ginv = 1.f;
    if (c->dtype == VAE_DTYPE_F16) {
        const int e = std::max(0, ilog2(B) + 2 * ilog2(H) - 4);
        c->gmul = ldexpf(1.f, e); c->ginv = ldexpf(1.f, -e);
    }
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
        // a workgroup covers 64 quads of 4 output pixels per pass; few workgroups: one f64 atomic per channel each
        const int grid = (int)std::min<long>((P / 4 + 63) / 64, c->knob_conv1_grid);
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
        c->lay[i - 1].act_ok = 0;
        if (train && i >= 2 && c->lay[i - 1].act && raw_wgrad_ok<T>(c, i) && will_pipe(c, a)) {   // materialise a_{i-1} for layer i's weight gradient
            a.stage_out = reinterpret_cast<T*>(c->lay[i - 1].act); c->lay[i - 1].act_ok = 1;
        }
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


// Fused input + weight gradient of a ConvTranspose2d layer with a 32-channel high-res side (conv_fused.cuh): layers 7
// (final_layer.0) and 6 (decoder.2).  Returns 1 when the shape / storage type is outside the fused kernel's domain (the
// caller then takes the separate kernels), 0 on success, -1 on error.
template <typename T>
static bool convt_fused_ok(vae_ctx* c, int i) {
    if (sizeof(T) != 2) return false;
    const BnLayer& l = c->lay[i]; const BnLayer& lp = c->lay[i - 1];
    const int Hs = l.H / 2, Ws = l.W / 2, CLO = lp.C;
    if (!(c->use_fused_wgrad & 1) || !c->use_pipelined || l.C != 32 || (CLO != 32 && CLO != 64) || Hs % 8 || Ws % 16) return false;
    if (4.0 * c->B * Hs * Ws * 32 * sizeof(T) >= 4294967296.0 || 1.0 * c->B * Hs * Ws * CLO * sizeof(T) >= 4294967296.0) return false;   // 32-bit byte offsets
    const int grid = std::min(c->B * (Ws / 16) * (Hs / 8), c->knob_fused_grid);
    return (size_t)grid * 9 * CLO * 32 <= c->fused_slab_floats;
}
// recomp (layer 7 only): dz of the layer was not stored by the output-conv backward; it is recomputed from dl_src * dl_scale
template <typename T>
static int launch_convt_fused(vae_ctx* c, int i, const float* params, float* grads, hipStream_t st, bool recomp = false,
                              const float* dl_src = nullptr, const float* dl_scale = nullptr) {
    if constexpr (sizeof(T) != 2) return 1;
    else {
        if (!convt_fused_ok<T>(c, i)) return 1;
        const BnLayer& l = c->lay[i]; const BnLayer& lp = c->lay[i - 1];
        const int Hs = l.H / 2, Ws = l.W / 2, CLO = lp.C;
        const int fs = i == 7 ? 0 : 1;
        ConvTFusedArgs<T> a; memset(&a, 0, sizeof(a));
        a.dz = reinterpret_cast<const T*>(l.dz); a.y = reinterpret_cast<const T*>(l.y); a.gcoef = l.block + LC_P0 * l.C;
        a.fuse = make_fuse_bwd(c, i, params, grads);
        if (!c->use_fused_bn) { if (bn_finalize_now(c, a.fuse, st)) return -1; a.fuse.mode = BNF_NONE; }   // standalone finalisation: coefficients from the block
        a.wp = reinterpret_cast<const T*>(c->wp_dg[i]);
        a.yprev = reinterpret_cast<const T*>(lp.y); a.ocoef = lp.block; a.dzprev = reinterpret_cast<T*>(lp.dz); a.stat = lp.stat_b;
        a.slab = c->fused_slab[fs]; a.slope = kSlope;
        a.B = c->B; a.Hs = Hs; a.Ws = Ws; a.tiles_x = Ws / 16; a.tiles_y = Hs / 8; a.n_tiles = c->B * a.tiles_x * a.tiles_y;
        a.rev = (c->knob_rev >> 2) & 1;
        a.dlogit = dl_src; a.gscale = dl_scale; a.gmul = c->gmul; a.wout = c->wout_t; a.fcoef = l.block;
        if (recomp && (CLO != 32 || !dl_src)) return vae_set_error("convt_fused", "recomputed dz: final_layer.0 only");
        const int grid = std::min(a.n_tiles, c->knob_fused_grid);
        const size_t lds = convt_fused_lds(CLO, recomp);
        const double px = (double)c->B * Hs * Ws;
        {
            ProfScope ps(c, recomp ? "convT_bwd_fused(dz recomputed+dgrad+wgrad)" : "convT_bwd_fused(dgrad+wgrad)",
                         sizeof(T) * ((recomp ? 1.0 : 2.0) * 4 * px * 32 + 2.0 * px * CLO + 9.0 * 32 * CLO) + 4.0 * 9 * 32 * CLO + (recomp ? 4.0 * 4 * px : 0.0),
                         2.0 * 2 * 9 * 32 * CLO * px + (recomp ? 2.0 * 9 * 32 * 4 * px : 0.0), st);
            if (recomp) { if (set_lds(convt_bwd_fused_kernel<T, 32, true>, lds)) return -1; hipLaunchKernelGGL((convt_bwd_fused_kernel<T, 32, true>), dim3(grid), dim3(512), lds, st, a); }
            else if (CLO == 32) { if (set_lds(convt_bwd_fused_kernel<T, 32>, lds)) return -1; hipLaunchKernelGGL((convt_bwd_fused_kernel<T, 32>), dim3(grid), dim3(512), lds, st, a); }
            else { if (set_lds(convt_bwd_fused_kernel<T, 64>, lds)) return -1; hipLaunchKernelGGL((convt_bwd_fused_kernel<T, 64>), dim3(grid), dim3(512), lds, st, a); }
            LAUNCH_CHECK("convt_bwd_fused_kernel");
        }
        // the per-workgroup slabs are summed beside the chain (the buffer is this layer's own: next written in the next step)
        SideFork f = fork_side(c, st);
        if (f.rc) return -1;
        if (launch_reduce(a.slab, grid, (size_t)9 * CLO * 32, grads + c->poff[l.p_convw], CLO, 32, f.st, c)) return -1;
        return 0;
    }
}

// Fused input + weight gradient of encoder.1 (Conv2d 32 -> 64 with the 32-channel tensor on the high-res side; conv_fused.cuh).
// Returns 1 when outside the fused kernel's domain (the caller takes the separate kernels), 0 on success, -1 on error.
template <typename T>
static int launch_conv_fused(vae_ctx* c, int i, const float* params, float* grads, hipStream_t st) {
    if constexpr (sizeof(T) != 2) return 1;
    else {
        const BnLayer& l = c->lay[i]; const BnLayer& lp = c->lay[i - 1];
        const int Hs = l.H, Ws = l.W;
        if (!(c->use_fused_wgrad & 2) || !c->use_pipelined || l.C != 64 || lp.C != 32 || Hs % 8 || Ws % 8) return 1;
        if (4.0 * c->B * Hs * Ws * 32 * sizeof(T) >= 4294967296.0 || 1.0 * c->B * Hs * Ws * 64 * sizeof(T) >= 4294967296.0) return 1;   // 32-bit byte offsets
        ConvFusedArgs<T> a; memset(&a, 0, sizeof(a));
        a.tiles_x = Ws / 8; a.tiles_y = Hs / 8; a.n_tiles = c->B * a.tiles_x * a.tiles_y;
        const int grid = std::min(a.n_tiles, c->knob_fused_grid);
        if ((size_t)grid * 9 * 64 * 32 > c->fused_slab_floats) return 1;
        a.dz = reinterpret_cast<const T*>(l.dz); a.y = reinterpret_cast<const T*>(l.y); a.gcoef = l.block + LC_P0 * l.C;
        a.fuse = make_fuse_bwd(c, i, params, grads);
        if (!c->use_fused_bn) { if (bn_finalize_now(c, a.fuse, st)) return -1; a.fuse.mode = BNF_NONE; }
        a.wp = reinterpret_cast<const T*>(c->wp_dg[i]);
        a.yprev = reinterpret_cast<const T*>(lp.y); a.ocoef = lp.block; a.dzprev = reinterpret_cast<T*>(lp.dz); a.stat = lp.stat_b;
        a.slab = c->fused_slab[2]; a.slope = kSlope; a.B = c->B; a.Hs = Hs; a.Ws = Ws; a.rev = (c->knob_rev >> 2) & 1; a.ablate = c->knob_ablate_f;
        const size_t lds = conv_fused_lds();
        const double px = (double)c->B * Hs * Ws;
        {
            ProfScope ps(c, "conv_bwd_fused(dgrad+wgrad)", sizeof(T) * (2.0 * px * 64 + 2.0 * 4 * px * 32 + 9.0 * 32 * 64) + 4.0 * 9 * 32 * 64,
                         2.0 * 2 * 9 * 32 * 64 * px, st);
            if (set_lds(conv_bwd_fused_kernel<T>, lds)) return -1;
            hipLaunchKernelGGL((conv_bwd_fused_kernel<T>), dim3(grid), dim3(512), lds, st, a);
            LAUNCH_CHECK("conv_bwd_fused_kernel");
        }
        SideFork f = fork_side(c, st);
        if (f.rc) return -1;
        if (launch_reduce(a.slab, grid, (size_t)9 * 64 * 32, grads + c->poff[l.p_convw], 64, 32, f.st, c)) return -1;
        return 0;
    }
}

template <typename T>
static int wgrad_on_side(vae_ctx* c, WgradArgs<T> w, float* dw_out, hipStream_t st, bool raw = false) {
    SideFork f = fork_side(c, st);
    if (f.rc) return f.rc;
    return launch_wgrad<T>(c, w, dw_out, f.st, f.slab, raw);
}
// can the weight gradient of BN layer i (2..5) run on materialised operands?  (16-bit storage, wide prefetching tile)
template <typename T>
static bool raw_wgrad_ok(vae_ctx* c, int i) {
    return sizeof(T) == 2 && c->use_raw_wgrad && c->use_pipelined && c->wk.wide && c->wk.tile == 1 && !c->wk.force_simple && i >= 2 && i <= 5;
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
    const bool step7 = c->convout_pending != 0;
    if (step7 && (g_xhat || gscale || !add_kl)) return vae_set_error("vae_backward", "the forward ran with train = 2: only the standard ELBO backward (no upstream gradient on xhat, no loss scale) can follow");
    if (!step7 && add_kl && !c->dlogit_valid) return vae_set_error("vae_backward", "this forward's output-conv gradient was already consumed (train = 2 forwards can be differentiated once)");
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
    bool recomp7 = false;
    {
        ConvOutBwdArgs a;
        a.yf = c->lay[7].y; a.ocoef = c->lay[7].block; a.wt = c->wout_t; a.dlogit = dl_src; a.gscale = dl_scale;
        a.dz = c->lay[7].dz; a.slab = c->use_side_stream ? c->side_slab[0] : c->slab; a.stat = c->lay[7].stat_b; a.dbias = c->accum + 2; a.B = B; a.H = H; a.W = H; a.slope = kSlope;
        a.gmul = c->gmul;   // the gradient scale enters the backward here (and in latent_bwd / fc_dgrad for the other upstream gradients)
        const long P = (long)B * H * H;
        int grid = (int)std::min<long>((P + 63) / 64, 1024);
        const bool will_recomp = sizeof(T) == 2 && c->use_mfma_convout && c->use_recomp_dz && convt_fused_ok<T>(c, 7);
        ProfScope ps(c, step7 ? "convout_step(fwd+bce+dgrad+wgrad)" : will_recomp ? "convout_bwd(statistics+wgrad, dz not stored)" : "convout_bwd(dgrad+wgrad+bn prologue)",
                     ((double)sizeof(T) * (will_recomp ? 32 : 64) + (step7 ? 8.0 : 4.0)) * P, (step7 ? 4.0 : 3.0) * 2 * 9 * 32 * P, st);
        bool launched = false;
        if constexpr (sizeof(T) == 2) {
            if (step7 && c->use_convout_stream && H == cos::RW) {
                // row-streaming form: units = (image, band of RB rows); bands only where whole images would leave CUs idle or the
                // last round mostly empty (a band costs RB/2 + 3 ticks and restages 4 rows)
                ConvOutStreamArgs<T> m; m.fuse = c->pending_f7;
                m.yf = reinterpret_cast<const T*>(c->lay[7].y); m.wt = c->wout_t; m.bias = params + c->poff[39]; m.target = c->x;
                m.xhat = c->xhat; m.accum = c->accum; m.dz = reinterpret_cast<T*>(c->lay[7].dz); m.slab = a.slab; m.stat = a.stat;
                m.B = B; m.H = H; m.inv_n = (float)(1.0 / ((double)B * H * H)); m.slope = kSlope; m.gmul = c->gmul;
                m.dbg = (c->dbg_buf && !strcmp(c->dbg_tag, "final_layer.3")) ? c->dbg_buf : nullptr;
                const int ncu = 256;
                long best = -1; int nb = 1;
                for (int cand = 1; cand <= 16 && H / cand >= 8; cand *= 2) {
                    const long rounds = ((long)B * cand + ncu - 1) / ncu, cost = rounds * (H / cand / 2 + 3);
                    if (best < 0 || cost < best) { best = cost; nb = cand; }
                }
                if (c->knob_convout_bands > 0 && H % c->knob_convout_bands == 0 && (H / c->knob_convout_bands) % 2 == 0 && H / c->knob_convout_bands >= 8) nb = c->knob_convout_bands;
                m.nb = nb; m.RB = H / nb; m.n_units = B * nb;
                grid = std::min(m.n_units, ncu);
                const size_t lds = convout_stream_lds();
                if (set_lds(convout_stream_kernel<T>, lds)) return -1;
                hipLaunchKernelGGL((convout_stream_kernel<T>), dim3(grid), dim3(1024), lds, st, m);
                launched = true;
                c->convout_pending = 0;
            }
            if (step7 && !launched) {
                ConvOutStepArgs<T> m; m.fuse = c->pending_f7; m.rev = (c->knob_rev >> 1) & 1;
                m.yf = reinterpret_cast<const T*>(c->lay[7].y); m.wt = c->wout_t; m.bias = params + c->poff[39]; m.target = c->x;
                m.xhat = c->xhat; m.accum = c->accum; m.dz = reinterpret_cast<T*>(c->lay[7].dz); m.slab = a.slab; m.stat = a.stat;
                m.B = B; m.H = H; m.W = H; m.n_tiles = B * (H / 8) * (H / 32);
                m.inv_n = (float)(1.0 / ((double)B * H * H)); m.slope = kSlope; m.gmul = c->gmul; m.ablate = c->knob_ablate_f;
                grid = std::min(m.n_tiles, c->knob_convout_step_grid);   // 512 resident (2 per CU by LDS): two full rounds
                const size_t lds = convout_step_lds();
                if (set_lds(convout_step_mfma_kernel<T>, lds)) return -1;
                hipLaunchKernelGGL((convout_step_mfma_kernel<T>), dim3(grid), dim3(256), lds, st, m);
                launched = true;
                c->convout_pending = 0;
            }
        }
        if constexpr (sizeof(T) == 2) {
            if (!launched && c->use_mfma_convout) {
                // final_layer.0's gradient kernel can recompute dz from dlogit: then this pass only produces the statistics and
                // the output conv's weight gradient, and the full-resolution 32-channel dz never goes to HBM
                recomp7 = c->use_recomp_dz && convt_fused_ok<T>(c, 7);
                ConvOutBwdMfmaArgs<T> m; m.rev = (c->knob_rev >> 1) & 1; m.store_dz = recomp7 ? 0 : 1;
                m.yf = reinterpret_cast<const T*>(c->lay[7].y); m.ocoef = a.ocoef; m.wt = a.wt; m.dlogit = a.dlogit; m.gscale = a.gscale; m.gmul = a.gmul;
                m.dz = reinterpret_cast<T*>(c->lay[7].dz); m.slab = a.slab; m.stat = a.stat; m.dbias = a.dbias;
                m.B = B; m.H = H; m.W = H; m.n_tiles = B * (H / 8) * (H / 32); m.slope = kSlope;
                grid = std::min(m.n_tiles, c->knob_convout_bwd_grid);
                hipLaunchKernelGGL((convout_bwd_mfma_kernel<T>), dim3(grid), dim3(256), 0, st, m);
                launched = true;
            }
        }
        if (!launched) hipLaunchKernelGGL((convout_bwd_kernel<T>), dim3(grid), dim3(256), 0, st, a);
        cgrid = grid;
        LAUNCH_CHECK("convout_bwd_kernel");
    }
    {
        SideFork f = fork_side(c, st, 0);   // the kernel above wrote its partial sums into side stream 0's slab
        if (f.rc) return f.rc;
        if (launch_reduce(f.slab, cgrid, 288, grads + c->poff[38], 1, 32, f.st, c)) return -1;
        hipLaunchKernelGGL(accum_to_f32_kernel, dim3(1), dim3(64), 0, f.st, c->accum + 2, grads + c->poff[39], c->ginv);
        LAUNCH_CHECK("accum_to_f32_kernel");
        if (step7 && c->loss_out3) {   // the ELBO scalars vae_loss_deferred asked for: the BCE sum exists only now
            hipLaunchKernelGGL(loss_finalize_kernel, dim3(1), dim3(64), 0, f.st, c->accum, c->loss_out3,
                               1.0 / ((double)B * H * H), 1.0 / (double)B, c->loss_kw, STAT_R);
            LAUNCH_CHECK("loss_finalize_kernel");
            c->loss_out3 = nullptr;
        }
    }
    // decoder stack: ConvTranspose2d layers 7 (final_layer.0), 6, 5, 4
    for (int i = 7; i >= 4; --i) {
        c->tag = kLayerTag[i];
        if (i >= 6) {   // 32-channel high-res side: one pass over (dz, y) for both gradients
            const int rc = launch_convt_fused<T>(c, i, params, grads, st, i == 7 && recomp7, dl_src, dl_scale);
            if (rc == 1 && i == 7 && recomp7) return vae_set_error("vae_backward", "dz of final_layer.0 was not stored");
            if (rc < 0) return -1;
            if (rc == 0) continue;
        }
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
        // deep layers: the input-gradient kernel materialises g = BN-backward(dz, y) while staging it; the weight gradient then
        // reads g and the forward's materialised activation as plain copies (it must follow the input-gradient launch)
        const bool raw = raw_wgrad_ok<T>(c, i) && l.dy && will_pipe(c, a) && (i == 4 || c->lay[i - 1].act_ok);
        if (raw) {
            a.stage_out = reinterpret_cast<T*>(l.dy);
            if (launch_down<T>(c, a, st)) return -1;
            w.s0 = i == 4 ? reinterpret_cast<const T*>(c->d0) : reinterpret_cast<const T*>(c->lay[i - 1].act); w.s1 = nullptr; w.s_two = 0; w.sslope = 1.f;
            w.g0 = reinterpret_cast<const T*>(l.dy); w.g1 = nullptr; w.g_two = 0; w.gslope = 1.f; w.fuse.mode = BNF_NONE;
            if (!((c->knob_skip_wgrad >> i) & 1) && wgrad_on_side<T>(c, w, grads + c->poff[l.p_convw], st, true)) return -1;
            continue;
        }
        if (!((c->knob_skip_wgrad >> i) & 1) && wgrad_on_side<T>(c, w, grads + c->poff[l.p_convw], st)) return -1;   // (knob: timing diagnostics)
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
        const bool lat_mfma = (c->use_latent_mfma & 4) && 2 * L <= 9 * 32;   // fc weight gradient path (also produces the bias column sums)
        if ((c->use_latent_mfma & 2) && L + 1 <= 9 * 32) {   // weight + bias gradient in one launch, no slabs (latent_mfma.cuh)
            SideFork f = fork_side(c, st);
            if (f.rc) return f.rc;
            BatchGemmArgs g; memset(&g, 0, sizeof(g));
            g.X = c->dd0; g.coef = nullptr; g.slope = 1.f; g.Y = c->z; g.ldy = L; g.ncols = L; g.ones_col = 1;
            g.out0 = grads + c->poff[20]; g.outb = grads + c->poff[21]; g.B = B; g.F = (int)c->F; g.L = L; g.s2 = c->s2; g.scale = c->ginv;
            ProfScope ps(c, "decin_wgrad", (double)sizeof(T) * B * (double)c->F + 4.0 * c->F * L, 2.0 * B * c->F * L, f.st);
            const size_t lds = batch_gemm_lds();
            if (set_lds(batch_gemm_kernel<T, false>, lds)) return -1;
            hipLaunchKernelGGL((batch_gemm_kernel<T, false>), dim3((unsigned)(c->F / 64)), dim3(256), lds, f.st, g);
            LAUNCH_CHECK("batch_gemm_kernel");
        } else {
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
        lb.gmu = g_mu; lb.glv = g_lv; lb.gz = g_z; lb.dlat = c->dlat; lb.B = B; lb.L = L; lb.kld_weight = kld_weight; lb.add_kl = add_kl; lb.gmul = c->gmul;
        hipLaunchKernelGGL(latent_bwd_kernel, dim3((B * L * LAT_LANES + 255) / 256), dim3(256), 0, st, lb);
        LAUNCH_CHECK("latent_bwd_kernel");
        if (!lat_mfma) {
            SideFork f = fork_side(c, st);
            if (f.rc) return f.rc;
            hipLaunchKernelGGL(colsum_kernel, dim3(2 * L), dim3(64), 0, f.st, c->dlat, B, 2 * L, grads + c->poff[17], grads + c->poff[19], L, c->ginv);
            LAUNCH_CHECK("colsum_kernel");
        }
    }
    // fc_mu / fc_var backward: weight gradients (side stream), then the input gradient with encoder.3's LeakyReLU / BatchNorm prologue
    if ((c->use_latent_mfma & 4) && 2 * L <= 9 * 32) {
        // both heads' weight gradients + their bias gradients (column sums of dlat) in one launch, no slabs (latent_mfma.cuh)
        SideFork f = fork_side(c, st);
        if (f.rc) return f.rc;
        BatchGemmArgs g; memset(&g, 0, sizeof(g));
        g.X = c->lay[3].y; g.coef = c->lay[3].block; g.slope = kSlope; g.Y = c->dlat; g.ldy = 2 * L; g.ncols = 2 * L; g.ones_col = 0;
        g.out0 = grads + c->poff[16]; g.out1 = grads + c->poff[18]; g.colsum0 = grads + c->poff[17]; g.colsum1 = grads + c->poff[19];
        g.B = B; g.F = (int)c->F; g.L = L; g.s2 = c->s2; g.scale = c->ginv;
        ProfScope ps(c, "fc_wgrad", (double)sizeof(T) * B * (double)c->F + 8.0 * c->F * L, 4.0 * B * c->F * L, f.st);
        const size_t lds = batch_gemm_lds();
        if (set_lds(batch_gemm_kernel<T, true>, lds)) return -1;
        hipLaunchKernelGGL((batch_gemm_kernel<T, true>), dim3((unsigned)(c->F / 64)), dim3(256), lds, f.st, g);
        LAUNCH_CHECK("batch_gemm_kernel");
    } else {
        FcWgradArgs<T> w;
        w.dlat = c->dlat; w.y = reinterpret_cast<const T*>(c->lay[3].y); w.coef = c->lay[3].block; w.slope = kSlope;
        w.B = B; w.F = (int)c->F; w.L = L; w.s2 = c->s2;
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
    if ((c->use_latent_mfma & 8) && 2 * L <= 256) {
        RowGemmArgs g; memset(&g, 0, sizeof(g));
        g.Y = c->dlat; g.ldy = 2 * L; g.K = 2 * L; g.Wp = c->fcpack; g.npad = c->npad_fc; g.out = c->lay[3].dz;
        g.y = c->lay[3].y; g.ocoef = c->lay[3].block; g.slope = kSlope; g.gpre = g_pre; g.gmul = c->gmul; g.stat = c->lay[3].stat_b;
        g.B = B; g.F = (int)c->F; g.s2 = c->s2;
        ProfScope ps2(c, "fc_dgrad", (double)sizeof(T) * (2.0 * B * c->F + 2.0 * c->F * L), 4.0 * B * c->F * L, st);
        const size_t lds = row_gemm_lds<T>();
        if (set_lds(row_gemm_kernel<T, 1>, lds)) return -1;
        hipLaunchKernelGGL((row_gemm_kernel<T, 1>), dim3((unsigned)(c->F / 64)), dim3(256), lds, st, g);
        LAUNCH_CHECK("row_gemm_kernel");
    } else {
        FcDgradArgs<T> d;
        d.dlat = c->dlat; d.wp = reinterpret_cast<const T*>(c->fcpack); d.npad = c->npad_fc; d.y = reinterpret_cast<const T*>(c->lay[3].y);
        d.ocoef = c->lay[3].block; d.slope = kSlope; d.gpre = g_pre; d.dz = reinterpret_cast<T*>(c->lay[3].dz); d.stat = c->lay[3].stat_b;
        d.B = B; d.F = (int)c->F; d.L2 = 2 * L; d.s2 = c->s2; d.gmul = c->gmul;
        ProfScope ps2(c, "fc_dgrad", (double)sizeof(T) * (2.0 * B * c->F + 2.0 * c->F * L), 4.0 * B * c->F * L, st);
        d.bt_per_wg = std::max(16, ((B + 7) / 8 + 15) / 16 * 16);   // <= 8 workgroups per channel: fewer same-address atomics
        bool wide = false;
        if constexpr (sizeof(T) == 2) wide = c->use_fc_dgrad8 && fc_dgrad8_lds(2 * L, 4, 16) <= 64 * 1024;
        if constexpr (sizeof(T) == 2) {
            if (wide) {
                // bit 1: 512-thread workgroups over 64 rows (half the workgroups, half the f64 atomics of the statistics)
                const bool big = (c->use_fc_dgrad8 & 2) && B > 32;
                d.bt_per_wg = big ? std::max(64, ((B + 3) / 4 + 63) / 64 * 64) : (d.bt_per_wg + 31) / 32 * 32;
                const dim3 grid((unsigned)(c->F / 256), (B + d.bt_per_wg - 1) / d.bt_per_wg);
                if (big) hipLaunchKernelGGL((fc_dgrad8_kernel<T, 4, 16>), grid, dim3(512), fc_dgrad8_lds(2 * L, 4, 16), st, d);
                else hipLaunchKernelGGL((fc_dgrad8_kernel<T, 4, 8>), grid, dim3(256), fc_dgrad8_lds(2 * L, 4, 8), st, d);
                LAUNCH_CHECK("fc_dgrad8_kernel");
            }
        }
        if (!wide) {
            hipLaunchKernelGGL((fc_dgrad_kernel<T>), dim3((unsigned)(c->F / 256), (B + d.bt_per_wg - 1) / d.bt_per_wg), dim3(256), 2 * L * 16 * 4, st, d);
            LAUNCH_CHECK("fc_dgrad_kernel");
        }
    }
    // encoder stack: Conv2d layers 3, 2, 1 on MFMA, then block 0
    for (int i = 3; i >= 1; --i) {
        c->tag = kLayerTag[i];
        if (i == 1) {   // 32-channel high-res side: one pass for both gradients
            const int rc = launch_conv_fused<T>(c, i, params, grads, st);
            if (rc < 0) return -1;
            if (rc == 0) continue;
        }
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
        const bool raw = raw_wgrad_ok<T>(c, i) && l.dy && will_pipe(c, a) && lp.act_ok;
        if (raw) {   // (as in the decoder loop)
            a.stage_out = reinterpret_cast<T*>(l.dy);
            if (launch_up<T>(c, a, st)) return -1;
            w.s0 = reinterpret_cast<const T*>(l.dy); w.s1 = nullptr; w.s_two = 0; w.sslope = 1.f;
            w.g0 = reinterpret_cast<const T*>(lp.act); w.g1 = nullptr; w.g_two = 0; w.gslope = 1.f; w.fuse.mode = BNF_NONE;
            if (!((c->knob_skip_wgrad >> i) & 1) && wgrad_on_side<T>(c, w, grads + c->poff[l.p_convw], st, true)) return -1;
            continue;
        }
        if (!((c->knob_skip_wgrad >> i) & 1) && wgrad_on_side<T>(c, w, grads + c->poff[l.p_convw], st)) return -1;   // (knob: timing diagnostics)
        if (launch_up<T>(c, a, st)) return -1;
    }
    {
        c->tag = kLayerTag[0];
        BnFuse fb0 = make_fuse_bwd(c, 0, params, grads);
        if (!c->use_fused_bn || !(c->knob_lean & 2)) { if (bn_finalize_now(c, fb0, st)) return -1; fb0.mode = BNF_NONE; }
        const long P = (long)B * (H / 2) * (H / 2);
        const int grid = (int)std::min<long>((P / 4 + 63) / 64, 512);   // (a thread takes quads of 4 output pixels)
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
int backward_impl(vae_ctx* c, const float* x, const float* params, float* grads, const float* g_xhat, const float* gscale,
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

template <typename T>
int pre_latents_impl(vae_ctx* c, float* out, hipStream_t st) {
    const long n = (long)c->B * c->F;
    hipLaunchKernelGGL((pre_latents_kernel<T>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st,
                       reinterpret_cast<const T*>(c->lay[3].y), c->lay[3].block, kSlope, out, c->B, (int)c->F, c->s2);
    LAUNCH_CHECK("pre_latents_kernel");
    return 0;
}
template <typename T>
int debug_tensor_impl(vae_ctx* c, const void* src, float* out, long n, int C, int HW, hipStream_t st) {
    hipLaunchKernelGGL((nhwc_to_nchw_kernel<T>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, reinterpret_cast<const T*>(src), out, n, C, HW);
    LAUNCH_CHECK("nhwc_to_nchw_kernel");
    return 0;
}
