"""MI355X-native mirror of the reference's step loop ``train.train_one_epoch`` (train.py:554-767)
plus the optimiser/scheduler construction it depends on (train.py:201-238).

Same signature, same order of operations (forward -> zero_grad -> loss -> backward ->
optimizer.step -> scheduler.step), same counters, console line and return tuple.  When the
model is this package's VanillaVAE, the criterion is ``model.loss`` and the optimiser is
``FusedAdamW``, each step is the fused HIP chain (no autograd graph); any other combination
still runs through the same kernels via autograd.  With ``torch.distributed`` initialised
(one process per GPU, RCCL) the optimised gradient ranges are all-reduced before the update
-- plain data parallelism, which the reference only prepares for (train.py:165-166,201,663).
"""
from __future__ import annotations

import os
import time
from collections import deque
from contextlib import nullcontext

import torch
import torch.distributed as dist

from .models import VanillaVAE
from .optim import FusedAdamW

BASE_BATCH_SIZE = 128  # train.py: lr_relative is quoted per 128 samples


def build_optimizer(config, model, steps_per_epoch: int):
    """train.py:201-238: lr = lr_relative * batch_size / 128; AdamW over the encoder and decoder
    groups only (fc_mu, fc_var, decoder_input, final_layer are never updated); OneCycleLR."""
    world = int(getattr(config, "world_size", 1))
    config.batch_size = config.batch_size_per_gpu * world
    config.lr = config.lr_relative * config.batch_size / BASE_BATCH_SIZE
    params = []
    if not getattr(config, "freeze_encoder", False):
        params.append({"params": model.encoder.parameters(), "lr": config.lr * getattr(config, "lr_encoder_mult", 1.0),
                       "name": "encoder"})
    params.append({"params": model.decoder.parameters(), "lr": config.lr * getattr(config, "lr_decoder_mult", 1.0),
                   "name": "decoder"})
    name = getattr(config, "optimizer", "AdamW")
    if name == "AdamW" and isinstance(model, VanillaVAE):
        optimizer = FusedAdamW(params, lr=config.lr, weight_decay=getattr(config, "weight_decay", 0.0))
    else:
        optimizer = getattr(torch.optim, name)(params, lr=config.lr, weight_decay=getattr(config, "weight_decay", 0.0))
    if getattr(config, "scheduler", "OneCycle").lower() != "onecycle":
        raise NotImplementedError(f"Scheduler {config.scheduler} not supported.")
    scheduler = torch.optim.lr_scheduler.OneCycleLR(
        optimizer, [p["lr"] for p in optimizer.param_groups], epochs=config.epochs, steps_per_epoch=steps_per_epoch)
    if _dist_active() and isinstance(model, VanillaVAE) and world > 1:
        sync_initial_state(model)            # every replica starts from rank 0's weights / BatchNorm buffers
        enable_library_allreduce(model)      # RCCL from inside the step library where every rank owns a GPU
    return optimizer, scheduler


def _dp_world():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def _dist_active() -> bool:
    return dist.is_available() and dist.is_initialized()


def sync_initial_state(model: VanillaVAE, src: int = 0):
    """Identical replicas before the first step: broadcast rank `src`'s flat parameters, BatchNorm running statistics and
    counters.  The reference never needs this (one process); data parallelism does - ranks that construct their
    model under different RNG states would otherwise average gradients of different networks."""
    if not _dist_active() or model._flat is None:
        return
    for t in (model._flat, model._bnflat, model._nbt):
        dist.broadcast(t, src=src)


def enable_library_allreduce(model: VanillaVAE) -> bool:
    """Ask the model to exchange gradients through the step library's own RCCL communicator (include/vae_step.h:
    vae_comm_init / vae_allreduce_grads) instead of torch.distributed's: the collective is then enqueued by the library on
    the stream it names, between its own kernels.  Only with the "nccl" (= RCCL) backend - every rank needs its own GPU.
    The communicator is created with the context (collectively, on every rank's first forward)."""
    if not _dist_active() or dist.get_backend() != "nccl" or os.environ.get("VAE_DP_LIBRARY_COMM", "1") == "0":
        return False
    model._want_lib_comm = True
    return True


def _allreduce_range(model: VanillaVAE, prefix: str):
    """torch.distributed fallback (gloo rehearsals, or RCCL through ProcessGroupNCCL when the library communicator is
    unavailable): in-line synchronous op, so torch enqueues it on the CURRENT stream (asynchronous ops go through
    ProcessGroupNCCL's own stream and two event hand-offs, measured at ~0.25 ms per collective on MI355X).  Leaves the
    MEAN over ranks in the gradient buffer."""
    g = model.flat_grads()
    off, n = model.group_range(prefix)
    world = _dp_world()
    if dist.get_backend() == "nccl":
        dist.all_reduce(g[off:off + n], op=dist.ReduceOp.AVG, async_op=False)
    else:
        dist.all_reduce(g[off:off + n], op=dist.ReduceOp.SUM, async_op=False)
        if world > 1:
            g[off:off + n].mul_(1.0 / world)


def _reduce(model: VanillaVAE, prefixes, on_comm_stream: bool = False):
    """Mean all-reduce of the named gradient ranges: the library's communicator when every rank has one, else torch's."""
    if model.library_comm_world() == _dp_world():
        model.allreduce_ranges(prefixes, on_comm_stream=on_comm_stream)
        return
    with (torch.cuda.stream(model.comm_stream()) if on_comm_stream else nullcontext()):
        for prefix in prefixes:
            _allreduce_range(model, prefix)


def allreduce_gradients(model: VanillaVAE, optimizer=None):
    """Average the optimised gradient ranges over ranks (RCCL all-reduce over xGMI) on the current stream: afterwards
    ``param.grad`` holds the mean over replicas, the reference's single-process semantics at the global batch (up to
    per-replica BatchNorm).  Returns [] (kept for callers that wait on handles)."""
    if _dist_active():
        _reduce(model, ("decoder", "encoder"))   # decoder gradients are produced first
    return []


def _refuse_overlap_on_eight_queues():
    """The overlapped (bucketed) gradient exchange with eight or more HIP hardware queues runs the whole step 2.5x slower on MI355X
    (2.8 ms against 1.1 ms; 1.18 ms with six queues - tools/diag/gpu_dp_exchange.py): refuse it loudly instead
    of training at that speed.  The queue count is read by HIP when it starts, so it cannot be corrected from here."""
    try:
        queues = int(os.environ.get("GPU_MAX_HW_QUEUES", "4"))
    except ValueError:
        queues = 4
    if queues >= 8:
        raise RuntimeError(f"overlapped gradient exchange (overlap=True / VAE_DP_OVERLAP=1) with GPU_MAX_HW_QUEUES={queues}: this "
                           "combination is 2.5x slower than the in-line exchange on MI355X; export GPU_MAX_HW_QUEUES=6 (or "
                           "VAE_DP_OVERLAP=1, which selects 6) before the process starts, or use the default in-line exchange")


def fused_step(model: VanillaVAE, optimizer, x, eps=None, use_device_eps: bool = True, overlap: bool | None = None):
    """One training step on the fused path (train.py:634-656): forward, ELBO, backward, [gradient all-reduce], AdamW.
    Data parallel (the product path on a multi-GPU node): by default both gradient buckets go out as ONE RCCL group
    through the library's communicator on the compute stream, between the last backward kernel and the AdamW kernel
    (3 MB of f32 gradients: measured +15 us per step with a one-rank communicator on MI355X).  ``overlap=True`` puts the
    decoder bucket's all-reduce on the context's communication stream between the two halves of the backward, under
    the encoder half, the encoder bucket following in line; measured here that costs more than the ~40 us it can hide
    (+70 us with six hardware queues: event hand-offs and a sixth stream; with eight queues the whole step runs 2.5x
    slower, which this function refuses), so it is opt-in: ``overlap=True`` or VAE_DP_OVERLAP=1."""
    one_call = isinstance(optimizer, FusedAdamW) and os.environ.get("VAE_ONE_CALL_STEP", "1") != "0"
    if one_call:
        optimizer._bind()
        one_call = optimizer._model is model and len(optimizer.param_groups) == len(optimizer._ranges)
    if not _dist_active():
        if one_call:   # the whole step in one library call (include/vae_step.h: vae_train_step_fused)
            return model.fused_train_step(optimizer, x, eps=eps, use_device_eps=use_device_eps, exchange=0)
        out3, xhat = model.fused_forward_backward(x, eps=eps, use_device_eps=use_device_eps)
    else:
        if overlap is None:
            overlap = os.environ.get("VAE_DP_OVERLAP", "0") == "1"
        if overlap:
            _refuse_overlap_on_eight_queues()
        if one_call:
            model._context(x.shape[0])   # (creates the context - and with it the library's communicator - on the first step)
        if one_call and model.library_comm_world() == _dp_world():
            # data parallel through the library's communicator: in-line group (1) or bucketed exchange (2) inside the same call
            return model.fused_train_step(optimizer, x, eps=eps, use_device_eps=use_device_eps, exchange=2 if overlap else 1)
        if overlap:
            out3, xhat = model.fused_forward_backward(x, eps=eps, use_device_eps=use_device_eps,
                                                      on_decoder_grads=lambda: _reduce(model, ("decoder",), on_comm_stream=True))
            _reduce(model, ("encoder",))
        else:
            out3, xhat = model.fused_forward_backward(x, eps=eps, use_device_eps=use_device_eps)
            _reduce(model, ("decoder", "encoder"))
    optimizer.step()
    return out3, xhat


def pack_bits(stimuli: torch.Tensor) -> torch.Tensor:
    """Host-side helper for dataloaders: a 0/1 pianoroll batch [B,1,H,W] (any dtype) as bit planes [B,1,H,W/8] uint8 (most
    significant bit first, numpy.packbits order) - 1/32 of the float32 bytes.  train_one_epoch expands such batches on the device."""
    b = (stimuli != 0).to(torch.uint8)
    if b.shape[-1] % 8:
        raise ValueError("pack_bits: the image width must be a multiple of 8")
    w = torch.tensor([128, 64, 32, 16, 8, 4, 2, 1], dtype=torch.uint8)
    return (b.reshape(*b.shape[:-1], b.shape[-1] // 8, 8) * w).sum(-1).to(torch.uint8)


def _expand_stimuli(stimuli: torch.Tensor, width: int, device=None) -> torch.Tensor:
    """Device-side expansion of byte (uint8 / bool cells) or bit-plane (uint8, last dimension width / 8) stimuli to the float32
    tensor the kernels read; float32 input passes through.  On a GPU this is one library kernel (include/vae_step.h:
    vae_expand_stimuli) which also reads PINNED host batches in place - no separate host-to-device copy in the step's chain;
    the caller keeps such a host batch alive until the stream has passed the kernel.  CPU tensors with no target device (and
    cell counts that are not a multiple of 8) take the torch expression of the same mapping."""
    if stimuli.dtype == torch.float32:
        return stimuli if device is None else stimuli.to(device)
    planes = stimuli.dtype == torch.uint8 and stimuli.shape[-1] * 8 == width
    target = torch.device(device) if device is not None else stimuli.device
    if target.type == "cuda" and stimuli.dtype in (torch.uint8, torch.bool):
        n_cells = stimuli.numel() * (8 if planes else 1)
        if not (stimuli.is_cuda or (stimuli.is_pinned() and stimuli.is_contiguous())):
            stimuli = stimuli.to(target)          # pageable host memory: the ordinary (blocking) copy first
        if n_cells % 8 == 0 and n_cells > 0:
            from . import _lib
            src = stimuli.contiguous()
            if target.index is None:
                target = torch.device("cuda", torch.cuda.current_device())
            out = torch.empty(*src.shape[:-1], width if planes else src.shape[-1], dtype=torch.float32, device=target)
            with torch.cuda.device(target):
                _lib.check(_lib.lib().vae_expand_stimuli(src.data_ptr(), 1 if planes else 0, out.data_ptr(), n_cells,
                                                         torch.cuda.current_stream(target).cuda_stream), "vae_expand_stimuli")
            return out
        stimuli = stimuli.to(target)
    elif device is not None:
        stimuli = stimuli.to(device)
    if planes:
        shifts = torch.arange(7, -1, -1, device=stimuli.device, dtype=torch.uint8)
        bits = (stimuli.unsqueeze(-1) >> shifts) & 1
        return bits.reshape(*stimuli.shape[:-1], width).to(torch.float32)
    return stimuli.to(torch.float32)


def train_one_epoch(config, model, optimizer, scheduler, criterion, dataloader, device="cuda", epoch=1, n_epoch=None,
                    total_step=0, n_samples_seen=0, verbose=False):
    """Train the model for one epoch (train.py:554-767).  Stimuli may also arrive as uint8 / bool pianorolls or as bit planes
    (``pack_bits``): the host-to-device copy shrinks 4x / 32x and the batch is expanded to float32 on the device.  (A copy
    stream that prefetches the next batch was measured too: 1.19-1.54 ms/step, erratic, against a steady 1.24 for the blocking
    4 MB copy - tools/diag/gpu_loop_cost.py.)"""
    model.train()
    log_wandb = bool(getattr(config, "log_wandb", False))
    if log_wandb:
        import wandb  # lazy, optional (train.py:608-610)
    loss_epoch_dev = None
    if getattr(config, "print_interval", None) is None:
        config.print_interval = config.log_interval
    world = _dp_world()
    fused = (isinstance(model, VanillaVAE) and isinstance(optimizer, FusedAdamW)
             and getattr(criterion, "__self__", None) is model and not getattr(config, "freeze_encoder", False))
    n_batches = len(dataloader)
    in_flight = deque()   # pinned host batches a device kernel may still be reading, with the event that follows that kernel
    for batch_idx, (stimuli, y_true) in enumerate(dataloader):
        batch_size_this_gpu = stimuli.shape[0]
        # float32 stimuli: the blocking copy of train.py:630-631 (measured on MI355X, asynchronous copies of the pinned 16.8 MB batch
        # are SLOWER here - 1.80 ms/step on the compute stream, 3.0 ms/step prefetched on a copy stream - than its 1.66 ms/step).
        # Byte / bit-plane stimuli: expanded by one library kernel, which reads a PINNED host batch in place (no copy command, no
        # host wait: a blocking copy makes the host wait for the previous step before it enqueues the next one, ~80 us of idle GPU
        # per step); the host batch is kept alive until the stream has passed that kernel.
        # (train.py:631 also copies y_true to the device; nothing in the loop reads the labels, and the blocking copy of an unpinned
        #  tensor is one more host-device rendezvous per step: left on the host)
        host_batch = stimuli if (stimuli.dtype != torch.float32 and stimuli.device.type == "cpu" and stimuli.is_pinned()) else None
        stimuli = _expand_stimuli(stimuli, getattr(model, "img_size", stimuli.shape[-1]), device=device)
        if host_batch is not None and stimuli.is_cuda:
            ev = torch.cuda.Event(); ev.record(torch.cuda.current_stream(stimuli.device))
            in_flight.append((host_batch, ev))
            while in_flight and in_flight[0][1].query():
                in_flight.popleft()
            if len(in_flight) > 8:
                in_flight.popleft()[1].synchronize()
        if fused:
            # train.py:634-656 as one HIP chain: forward, ELBO, backward, [all-reduce], AdamW
            out3, reconstruction = fused_step(model, optimizer, stimuli, use_device_eps=False)
        else:
            with torch.no_grad() if getattr(config, "freeze_encoder", False) else nullcontext():
                output = model.forward(stimuli)
                reconstruction = output["output"]
            optimizer.zero_grad()
            loss_output = criterion(output)
            loss_output["loss"].backward()
            if isinstance(model, VanillaVAE):
                allreduce_gradients(model, optimizer)
            optimizer.step()
            out3 = torch.stack([loss_output["loss"].detach(), loss_output["reconstruction_loss"].detach(),
                                loss_output["kld_loss"].detach()])
        scheduler.step()
        total_step += 1
        batch_size_all = batch_size_this_gpu * getattr(config, "world_size", world)
        n_samples_seen += batch_size_all
        # The reference reads the three scalars with .item() every step (train.py:672-674), i.e. one host-device round trip
        # per step.  Here they stay on the device unless this step prints or logs them: the epoch sum is accumulated on the
        # device in float64, in step order - bit for bit the Python sum of the per-step float32 values.
        rank0 = getattr(config, "global_rank", 0) == 0
        printing = batch_idx <= 2 or batch_idx % config.print_interval == 0 or batch_idx >= n_batches - 1
        logging = log_wandb and rank0 and batch_idx % config.log_interval == 0
        first_verbose = epoch <= 1 and batch_idx == 0 and verbose
        if loss_epoch_dev is None:
            loss_epoch_dev = torch.zeros((), dtype=torch.float64, device=out3.device)
        loss_epoch_dev.add_(out3[0])          # (float64 += float32 in one kernel: the operand is widened inside it)
        if (printing and rank0) or logging or first_verbose:
            loss_batch, loss_recon, loss_kld = out3.tolist()  # one D2H sync for the three .item() of train.py:672-674
        if first_verbose:
            print("stimuli.shape =", stimuli.shape)
            print("logits.shape  =", reconstruction.shape)
            print("loss =", loss_batch)
        if printing:
            if rank0:
                print(
                    f"Train Epoch:{epoch:4d}" + (f"/{n_epoch}" if n_epoch is not None else ""),
                    f" Step:{batch_idx + 1:4d}/{n_batches}",
                    f" Loss:[F: {loss_batch:6.3f}, KL: {loss_kld:6.3f}]",
                    f" LR: {scheduler.get_last_lr()[0]:.5f}",
                    f" KL Weight: {model.kld_weight:.5f}",
                )
        if logging:
            wandb.log({
                "training/stepwise/epoch": epoch,
                "training/stepwise/epoch_progress": epoch - 1 + (batch_idx + 1) / n_batches,
                "training/stepwise/n_samples_seen": n_samples_seen,
                "training/stepwise/train/loss": loss_batch,
                "training/stepwise/train/loss_recon": loss_recon,
                "training/stepwise/train/loss_kld": loss_kld,
                "training/stepwise/train/kld_weight": model.kld_weight,
            }, step=total_step)
    loss_epoch = float(loss_epoch_dev) if loss_epoch_dev is not None else 0.0   # the epoch's one unconditional synchronisation
    results = {"loss": loss_epoch / n_batches}
    return results, total_step, n_samples_seen


class SyntheticPianorollLoader:
    """Batches of the synthetic line/pianoroll distribution (data_generators.py:45-77), generated on the
    device by the library (include/vae_step.h: vae_synth_pianoroll).  Yields (stimuli, y_true)."""

    def __init__(self, batch_size: int, img_size: int, n_batches: int, seed: int = 0, device="cuda", pool: int = 0):
        self.batch_size, self.img_size, self.n_batches, self.seed, self.device = batch_size, img_size, n_batches, seed, device
        self.pool = pool
        self._cache = {}

    def __len__(self):
        return self.n_batches

    def batch(self, i: int):
        from . import _lib
        key = i % self.pool if self.pool else None
        if key is not None and key in self._cache:
            return self._cache[key]
        x = torch.empty(self.batch_size, 1, self.img_size, self.img_size, device=self.device, dtype=torch.float32)
        _lib.check(_lib.lib().vae_synth_pianoroll(x.data_ptr(), self.batch_size, self.img_size,
                                                  self.seed + (key if key is not None else i),
                                                  torch.cuda.current_stream().cuda_stream), "vae_synth_pianoroll")
        y = torch.zeros(self.batch_size, dtype=torch.long, device=self.device)
        if key is not None:
            self._cache[key] = (x, y)
        return x, y

    def __iter__(self):
        for i in range(self.n_batches):
            yield self.batch(i)
