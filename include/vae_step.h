/* C ABI of the MI355X-native VanillaVAE training step (libvae_step_gfx950.so).
 *
 * The reference has no FFI / plugin interface: its hot path is the
 * Python call surface models.VanillaVAE.{forward,loss} + train.train_one_epoch,
 * dispatching to PyTorch ATen.  This library sits BENEATH that surface; each entry
 * point names the reference lines whose arithmetic it replaces.  Paths are relative
 * to /root/reference/midi_autoencoder.
 *
 * Conventions: plain pointers and sizes, no torch types.  All tensor memory
 * (parameters, gradients, optimiser state, inputs, outputs) is owned by the caller
 * (PyTorch's allocator) and borrowed for the call; only scratch is owned by the
 * context.  Work is ordered on the explicit hipStream_t (pass
 * torch.cuda.current_stream().cuda_stream): results of a call are visible to later
 * work on that stream, and a call sees everything enqueued on it before.  Inside
 * vae_forward / vae_backward the context also uses non-blocking side streams of its own
 * (weight packing, weight gradients); they are forked from and joined back into the
 * caller's stream with HIP events before the call returns, so the caller never has to
 * synchronise with them.  Functions return
 * 0 on success or a negative code, with the message in vae_last_error().  A context is
 * not re-entrant; use one per process / GPU.
 */
#ifndef VAE_STEP_H
#define VAE_STEP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct vae_ctx vae_ctx;
typedef void* vae_stream_t; /* hipStream_t */

#define VAE_NUM_PARAMS 40 /* tensors of VanillaVAE.state_dict() that are parameters */
#define VAE_NUM_BN 8
#define VAE_DTYPE_F32 0  /* f32 storage, f32-input MFMA: exact f32 FMA-chain arithmetic */
#define VAE_DTYPE_BF16 1 /* bf16 activation/weight storage, bf16 MFMA, f32 accumulate/statistics */
#define VAE_DTYPE_F16 2  /* f16 activation/weight storage, f16 MFMA, f32 accumulate; KL / BCE / BatchNorm statistics in
                          * f32 / f64 as in the other modes.  Stored gradients are scaled by a power of two chosen per
                          * forward (the BCE mean makes dL/dlogit ~ 1/(B*H*W), below the f16 range) and every parameter
                          * gradient is unscaled on output, so callers see ordinary gradients. */

const char* vae_last_error(void);
int vae_abi_version(void);

/* Flat parameter buffer layout.  Tensors appear in the reference's state_dict order
 * (models.py:41-82): encoder.{0..3}.{0.weight,0.bias,1.weight,1.bias}, fc_mu.{weight,bias},
 * fc_var.{weight,bias}, decoder_input.{weight,bias}, decoder.{0..2}.{...}, final_layer.{0.weight,
 * 0.bias,1.weight,1.bias,3.weight,3.bias}; each keeps the reference's own element layout
 * (Conv2d [Cout,Cin,3,3], ConvTranspose2d [Cin,Cout,3,3], Linear [out,in]).
 * generalised=0: flattened_size = 1024 (models.py:33,36; img_size must be 32).
 * generalised=1: flattened_size = 256*(img_size/16)^2 (not reference behaviour). */
int vae_param_layout(int img_size, int latent_dim, int generalised, int64_t* offsets /*[40]*/,
                     int64_t* sizes /*[40]*/, int64_t* total);
/* BatchNorm running statistics: one f32 buffer, per layer running_mean[C] then running_var[C]. */
int vae_bn_layout(int64_t* offsets /*[8]*/, int64_t* channels /*[8]*/, int64_t* total);

/* Replaces VanillaVAE.__init__'s device state (models.py:10-83): allocates scratch for
 * batches up to max_batch.  dtype: VAE_DTYPE_*. */
vae_ctx* vae_create(int img_size, int latent_dim, int max_batch, int dtype, int generalised);
void vae_destroy(vae_ctx* ctx);
/* bytes of device scratch held by the context */
int64_t vae_workspace_bytes(const vae_ctx* ctx);

/* VanillaVAE.forward (models.py:185-188): encode (:107-145), reparameterize (:177-183),
 * decode (:147-175); also accumulates the ELBO terms of VanillaVAE.loss (:208,:214) and the
 * reconstruction gradient so that vae_loss / vae_backward need no second pass.
 *   x [B,1,H,W] f32 in [0,1];  params: flat buffer (vae_param_layout)
 *   bn_running / num_batches_tracked[8]: updated when train!=0 (momentum 0.1, unbiased var)
 *   eps [B,L] f32: the torch.randn_like draw of models.py:182; NULL -> generated on device
 *       from the counter-based normal generator (seed, stream 5) that oracle/ restates
 *   train=0 uses running statistics (model.eval(), evaluation.py:42)
 *   train=2: a training forward whose standard-ELBO backward follows unconditionally (the fused step, train.py:634-650
 *       as one chain).  The library MAY then leave the output conv, sigmoid and BCE to vae_backward / vae_backward_part,
 *       where one kernel does that layer's forward and backward in a single pass over its input: xhat, the running
 *       statistics of final_layer's BatchNorm and the ELBO scalars (through vae_loss_deferred, which must be used
 *       instead of vae_loss) are then written by the backward, which must be the standard one (use_std = 1, no
 *       upstream gradient on xhat, no loss scale) and can run once.  x and xhat must stay valid until it has run.
 *   outputs: xhat [B,1,H,W], mu/log_var/z [B,L], all f32. */
int vae_forward(vae_ctx* ctx, const float* x, int batch, const float* params, float* bn_running,
                int64_t* num_batches_tracked, const float* eps, uint64_t seed, int train, float* xhat,
                float* mu, float* log_var, float* z, vae_stream_t stream);

/* VanillaVAE.decode (models.py:147-175): z [B,L] -> xhat [B,1,H,W].  train!=0 uses (and updates) batch
 * statistics like a train-mode module call; the pass is not differentiable (vae_backward needs vae_forward). */
int vae_decode(vae_ctx* ctx, const float* z, int batch, const float* params, float* bn_running,
               int64_t* num_batches_tracked, int train, float* xhat, vae_stream_t stream);

/* EncoderOutput.pre_latents (models.py:133, types_helpers.py:20) of the last forward,
 * [B, flattened_size] f32 in the reference's NCHW-flatten order. */
int vae_pre_latents(vae_ctx* ctx, float* out, vae_stream_t stream);
/* eps actually used by the last forward, [B,L]. */
int vae_last_eps(vae_ctx* ctx, float* out, vae_stream_t stream);

/* VanillaVAE.loss (models.py:190-225) for the last forward: out3 = {loss, reconstruction_loss,
 * kld_loss} with kld_loss sign-flipped as at models.py:224. */
int vae_loss(vae_ctx* ctx, float kld_weight, float* out3, vae_stream_t stream);
/* The same scalars computed beside the backward instead of in front of it (one launch less on the critical chain):
 * enqueued on a context side stream ordered after `stream`; out3 is ordered into the caller's stream by the NEXT
 * vae_backward / vae_backward_part on this context, which must follow (train-mode forward only).  For callers that read
 * the ELBO after the step, as train_one_epoch does (train.py:644-674). */
int vae_loss_deferred(vae_ctx* ctx, float kld_weight, float* out3, vae_stream_t stream);

/* VanillaVAE.loss (models.py:190-225) on arbitrary caller tensors: xhat/target [n], mu/log_var
 * [B,L].  Optional outputs (NULL to skip): unscaled gradients of the loss w.r.t. xhat, mu, log_var
 * (BCE grad (x-t)/max(x(1-x),1e-12)/n as ATen computes it). */
int vae_elbo_generic(const float* xhat, const float* target, const float* mu, const float* log_var, int64_t n,
                     int batch, int latent_dim, float kld_weight, float* out3, float* g_xhat, float* g_mu,
                     float* g_log_var, vae_stream_t stream);

/* loss.backward() (train.py:650) for the last train-mode forward.
 *   grads: flat f32 buffer, same layout as params; every tensor is overwritten.
 *   use_std: 1 adds the gradient of the standard ELBO of vae_loss (reconstruction term fused in
 *           the forward, plus d(kld_weight*KL)/d(mu,log_var), models.py:208-216), scaled by
 *   gscale: device scalar = upstream gradient of loss.backward(), NULL = 1.
 *   g_xhat [B,1,H,W], g_mu/g_log_var/g_z [B,L], g_pre [B,F]: optional additional upstream
 *           gradients on the ModelOutput tensors (NULL = none). */
int vae_backward(vae_ctx* ctx, const float* x, const float* params, float* grads, const float* g_xhat,
                 const float* gscale, const float* g_mu, const float* g_log_var, const float* g_z,
                 const float* g_pre, float kld_weight, int use_std, vae_stream_t stream);
/* The same backward in two halves, for data-parallel callers that start the all-reduce of the decoder
 * gradients while the encoder half still runs: part 1 = output conv + decoder stack (on return every decoder
 * and final_layer gradient is complete in stream order), part 2 = the rest (decoder_input, latent, fc, encoder);
 * part 0 = both (= vae_backward).  Same arguments in both calls. */
int vae_backward_part(vae_ctx* ctx, const float* x, const float* params, float* grads, const float* g_xhat,
                 const float* gscale, const float* g_mu, const float* g_log_var, const float* g_z,
                 const float* g_pre, float kld_weight, int use_std, int part, vae_stream_t stream);
/* A non-blocking stream owned by the context, ordered after everything enqueued on `stream` so far.  Data-parallel
 * callers enqueue the decoder bucket's all-reduce on it right after part 1; it is joined back into the caller's
 * stream at the end of part 2, so the collective overlaps the encoder half without any host-side handshake. */
int vae_comm_stream(vae_ctx* ctx, vae_stream_t stream, vae_stream_t* out);

/* ---- data parallel: one process per GPU, gradients exchanged over RCCL (xGMI inside a node) -------------------------
 * The reference has no exchange step: it only scales the learning rate and the sample counters by
 * WORLD_SIZE (train.py:165-166, 201, 663).  These entry points are the gradient all-reduce north_star asks for, issued
 * by the library itself so that backward kernels, collective and AdamW share streams without a framework hand-off.
 * vae_comm_unique_id: rank 0 fills a VAE_COMM_ID_BYTES id; the host distributes it over any channel it has
 * (torch.distributed's store / broadcast, MPI, a file); every rank then calls vae_comm_init with the device it will run on
 * current.  The communicator belongs to the context (released by vae_comm_destroy / vae_destroy). */
#define VAE_COMM_ID_BYTES 128
int vae_comm_unique_id(void* id /*[VAE_COMM_ID_BYTES]*/);
int vae_comm_init(vae_ctx* ctx, int rank, int world, const void* id /*[VAE_COMM_ID_BYTES]*/);
int vae_comm_world(const vae_ctx* ctx); /* world size of the context's communicator, 0 if none */
int vae_comm_destroy(vae_ctx* ctx);
/* In-place all-reduce over ranks of `nranges` ranges [offsets[i], offsets[i]+sizes[i]) of the flat f32 gradient buffer,
 * as one RCCL group enqueued on `stream`: the caller's stream (ordered after the backward by the stream itself) or the
 * context's communication stream (vae_comm_stream) for the decoder bucket between vae_backward_part 1 and 2.
 * average != 0 leaves the MEAN over ranks (what a data-parallel caller expects in .grad); 0 the sum. */
int vae_allreduce_grads(vae_ctx* ctx, float* grads, int nranges, const int64_t* offsets, const int64_t* sizes, int average,
                        vae_stream_t stream);
/* Identical replicas before the first step: broadcast rank `root`'s flat parameters, BatchNorm running statistics and
 * num_batches_tracked (NULL to skip the latter two) over the context's communicator. */
int vae_broadcast_state(vae_ctx* ctx, float* params, float* bn_running, int64_t* num_batches_tracked, int root,
                        vae_stream_t stream);

/* torch.optim.AdamW.step (train.py:228,656) on up to two contiguous ranges of the flat
 * buffers (the encoder and decoder groups of train.py:210-225), each with the lr and beta1
 * OneCycleLR set for this step (train.py:233-238,659).  step is 1-based.  The hyper-parameters are
 * doubles, as torch holds them: 1 - beta, 1 - lr * weight_decay, the bias corrections and lr / bc1 are
 * formed in double and only then rounded to the float the element-wise update uses (torch's scalar
 * arguments: (float)(1 - 0.999) is 0.001f, not 1.f - 0.999f). */
int vae_adamw_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int ngroups,
                   const int64_t* offsets, const int64_t* sizes, const double* lrs, const double* beta1s,
                   double beta2, double eps, double weight_decay, float grad_scale, int step,
                   vae_stream_t stream);

/* One whole training step (train.py:634-659 minus logging): forward, loss, backward, AdamW. */
int vae_train_step(vae_ctx* ctx, const float* x, int batch, float* params, float* grads, float* exp_avg,
                   float* exp_avg_sq, float* bn_running, int64_t* num_batches_tracked, const float* eps,
                   uint64_t seed, float kld_weight, int ngroups, const int64_t* offsets,
                   const int64_t* sizes, const double* lrs, const double* beta1s, double beta2, double adam_eps,
                   double weight_decay, int step, float* xhat, float* mu, float* log_var, float* z,
                   float* out3, vae_stream_t stream);

/* The fused training step of torch_vae_amd.train.fused_step as ONE host call (train.py:634-659): forward with the output
 * conv deferred to the backward, ELBO scalars, backward, gradient exchange, AdamW.  exchange: 0 none; 1 one RCCL group over
 * the optimised ranges between the backward and AdamW; 2 bucketed - the last range (decoder) is all-reduced on the context's
 * communication stream under the encoder half of the backward, the others after it, and each range's AdamW launch waits only
 * for its own bucket, so one group's update runs while the other's all-reduce is in flight (the data-parallel layout
 * train.py:165-166,201,663 prepare).  1 and 2 need vae_comm_init and produce bit-identical results. */
int vae_train_step_fused(vae_ctx* ctx, const float* x, int batch, float* params, float* grads, float* exp_avg,
                         float* exp_avg_sq, float* bn_running, int64_t* num_batches_tracked, const float* eps,
                         uint64_t seed, float kld_weight, int ngroups, const int64_t* offsets,
                         const int64_t* sizes, const double* lrs, const double* beta1s, double beta2, double adam_eps,
                         double weight_decay, float grad_scale, int step, int exchange, float* xhat, float* mu,
                         float* log_var, float* z, float* out3, vae_stream_t stream);

/* Synthetic pianoroll/line batch with the distribution of data_generators.py:45-77
 * (called as at :97-104), seeded; x [B,1,H,H] f32 in {0,1}.  Device-side generator. */
int vae_synth_pianoroll(float* x, int batch, int img_size, uint64_t seed, vae_stream_t stream);

/* Byte or bit-plane stimuli expanded to the float32 batch [B,1,H,W] the step reads (the host side of
 * train.py:630-631: the reference copies float32 stimuli; a 0/1 pianoroll needs 1/4 or 1/32 of those
 * bytes on the host link).  kind 0: one byte per cell, v -> (float)v; kind 1: bit planes, most
 * significant bit first (numpy.packbits order).  `src` is device memory or PINNED host memory (read in
 * place by the kernel; the caller keeps it alive and unchanged until the stream has passed this call);
 * pageable host memory is refused.  n_cells = B*H*W, a multiple of 8. */
int vae_expand_stimuli(const void* src, int kind, float* dst, int64_t n_cells, vae_stream_t stream);

/* Per-kernel timing for bench.py's roofline line: when enabled every launch of the step is
 * bracketed by HIP events on the launch stream; the report is a JSON array with, per kernel name,
 * calls, total ms, and total ALGORITHMIC bytes / flops (operand tensors once). */
int vae_profile(vae_ctx* ctx, int enable);
int vae_profile_report(vae_ctx* ctx, char* buf, int64_t capacity);
/* JSON array of the labels of all profiled launches, in launch order. */
int vae_profile_sequence(vae_ctx* ctx, char* buf, int64_t capacity);
/* JSON array [[label, start_ms, end_ms, algorithmic_bytes], ...] of all profiled launches, times relative to
 * the first one: weight gradients run on the context's side streams, so launches overlap. */
int vae_profile_timeline(vae_ctx* ctx, char* buf, int64_t capacity);
/* Diagnostics: per-wave phase cycle counters of the pipelined conv kernel of one layer.  tag = layer label
 * ("final_layer.0" ...), epi = epilogue kind (0 forward, 1 backward, 2 plain; +16 selects the transposed-conv
 * kernel), out = device buffer of grid*waves*8 int64 for the stride-2 conv kernel (6 loop phases + prologue + tail),
 * grid*4*6 for the transposed-conv kernel (NULL switches it off).  Needs a `make STAMPS=1` build. */
int vae_debug_stamps(vae_ctx* ctx, const char* tag, int epi, long long* out);

/* Debug / test hooks: copy an internal NHWC tensor to f32 NCHW.  which: 0..7 raw conv output
 * of BN layer i, 8..15 its dz, 16 decoder_input output, 17 its gradient. */
int vae_debug_tensor(vae_ctx* ctx, int which, float* out, int64_t capacity, vae_stream_t stream);
/* hardware self-test of the transposed LDS read used by the bf16 weight-gradient kernel */
int vae_selftest_tr16(vae_stream_t stream);
/* Tuning / diagnostic switches (defaults in brackets):
 *   use_tr16 [1]            ds_read_b64_tr_b16 in the bf16 weight-gradient kernel
 *   use_mfma_convout [1]    MFMA versions of the output-conv kernels (bf16)
 *   use_pipelined [1]       persistent prefetching conv kernels (0: one tile per workgroup)
 *   use_side_stream [1]     weight gradients / weight packing on the context's side streams
 *   use_fused_bn [1]        BatchNorm finalisation inside the consumer kernel's prologue
 *   use_fused_convout [1]   honour train = 2 (output conv forward + backward as one kernel; 16-bit storage);
 *                           knob_convout_step_grid [1024] its persistent workgroups (swept 512-4096: 1.328 / 1.323 / 1.334 / 1.341 / 1.354 ms)
 *   use_fused_wgrad [3]     bit 0: one pass over (dz, y) for the input AND weight gradient of final_layer.0 / decoder.2 / encoder.1
 *                           (16-bit storage; conv_fused.cuh): bit 0 the transposed-conv layers final_layer.0 / decoder.2, bit 1
 *                           encoder.1; knob_fused_grid [256] their persistent workgroups; use_recomp_dz [0] final_layer.0's dz
 *                           recomputed from dlogit instead of stored (bit-identical, measured slower);
 *                           use_raw_wgrad [0] deep weight gradients from operands materialised by the input-gradient kernels
 *                           (bit-identical, measured 1 % slower)
 *   knob_down_waves [8]     waves of the wide stride-2 conv kernels: 8 = 2x4 wave grid on 128-channel tiles and (knob_lay42 [1]) 4x2 on
 *                           64-channel tiles, 4 = 2x2
 *   knob_up_nt_max [1]      output channels per workgroup tile of the transposed-conv kernels in 32-channel blocks (1: more, smaller
 *                           workgroups at two waves per SIMD; 2 = one wave per SIMD measured 2.5 % slower on the step)
 *   knob_lay22_min_nt [2]   wave-grid layouts (16-bit storage) for output tiles of at least this many 32-channel blocks
 *   knob_rev [4]            reverse tile walk (bit 0 output-conv forward, 1 output-conv backward, 2 backward conv kernels,
 *                           3 weight-gradient kernels, 4 forward conv kernels, 5 alternate per launch): a consumer that starts with
 *                           what its producer wrote last finds it in L2 / the memory-side cache
 *   knob_wgrad_mid8 [0]     eight waves on the 64x32-channel weight-gradient tile (measured slower; diagnostics)
 *   knob_wgrad_force_simple [0]  take the 64-bit-offset weight-gradient kernel (the fallback for tensors >= 4 GiB) at any size
 *   knob_lean [1]           launches kept off the critical chain (bit 0 reparameterisation noise drawn beside the first conv,
 *                           1 BatchNorm backward of encoder block 0 inside its weight-gradient kernel, 2 vae_loss_deferred
 *                           really on a side stream)
 *   knob_wave_nt_max [4]    wave-independent tiles for output tiles of up to this many 32-channel blocks
 *   knob_nt_max [4], knob_up_per_cu [2] (resident workgroups per CU of the transposed-conv kernels: 2 beats 4 by 2 % of the step, 1 and 3 are worse), knob_down_per_cu [2] (measured flat 1-4), knob_convout_grid [1536], knob_convout_bwd_grid [1024] (persistent workgroups of the output-conv forward / backward kernels; full rounds of what is resident - 768 / 512 - beat 2048 by 1 %; the backward grid equals knob_convout_step_grid so that both partition the tiles alike), knob_pipe_max_cout [256], knob_bwd_per_cu [0],
 *   knob_wgrad_tile [1], knob_wgrad_wide [1], knob_wgrad_wgs [128], knob_wgrad_wide_wgs [128], knob_wgrad_cap_mb [48],
 *   knob_conv1_grid [1024], knob_pack_grid [128], knob_ablate_b [0]   grid / tile sizing
 *   knob_wgrad_layer_wgs [0]  diagnostic: (layer mask << 16) | workgroups overrides the weight-gradient split of the masked layers
 *   knob_skip_wgrad [0], knob_ablate_f [0]      ablation diagnostics (skip weight-gradient launches by layer mask / phases
 *                           of the encoder.1 fused kernel): results are WRONG when set - timing experiments only */
int vae_set_option(vae_ctx* ctx, const char* name, int value);

#ifdef __cplusplus
}
#endif
#endif
