"""MI355X-native mirror of the reference's ``models.VanillaVAE`` (models.py:7-272).

Same class name, constructor signature, attributes (``encoder``, ``decoder``,
``fc_mu``, ``fc_var``, ``decoder_input``, ``final_layer``, ``kld_weight``,
``latent_dim``), ``state_dict`` keys and output dict shapes, so code written
against the reference (``train.run``, ``evaluation.evaluate``) can use it
unchanged.  The arithmetic is not PyTorch's: ``forward``/``loss``/``backward``
call the hand-written gfx950 kernels behind ``include/vae_step.h``.

The torch sub-modules below are parameter CONTAINERS only (they give the
reference's names, shapes and initialisation); they are never called.  All
parameters are views into one flat f32 HBM buffer so the kernels, the fused
AdamW and the gradient all-reduce see single contiguous ranges.

Deviations from the reference, all explicit:
  * ``in_channels`` must be 1 and ``hidden_dims`` None/[32,64,128,256] (the only
    configuration the hot path in BASELINE.json names).
  * ``generalised=True`` (or ``input_dim != 32``) sizes the bottleneck as
    256*(input_dim/16)^2 instead of the hard-wired 1024 (models.py:33,166).
  * gradients are written to ``param.grad`` by the backward kernel directly
    (no AccumulateGrad hooks fire).
"""
from __future__ import annotations

import math
import contextlib
import weakref

import torch
from torch import Tensor, nn

from . import _lib
from .types_helpers import EncoderOutput, LossOutput, ModelOutput

_DTYPES = {"f32": _lib.DTYPE_F32, "fp32": _lib.DTYPE_F32, "float32": _lib.DTYPE_F32,
           "bf16": _lib.DTYPE_BF16, "bfloat16": _lib.DTYPE_BF16,
           "f16": _lib.DTYPE_F16, "fp16": _lib.DTYPE_F16, "float16": _lib.DTYPE_F16, "half": _lib.DTYPE_F16}


_NULL_GUARD = contextlib.nullcontext()


def _stream_ptr(device=None):
    return torch.cuda.current_stream(device).cuda_stream


def _rank() -> int:
    import torch.distributed as dist
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


class _Context:
    """Owns one vae_ctx (scratch sized for max_batch)."""

    def __init__(self, img_size, latent_dim, max_batch, dtype, generalised):
        self.key = (img_size, latent_dim, max_batch, dtype, generalised)
        self.handle = _lib.lib().vae_create(img_size, latent_dim, max_batch, dtype, int(generalised))
        if not self.handle:
            raise _lib.VaeLibError("vae_create: " + _lib.lib().vae_last_error().decode())

    def __del__(self):
        try:
            if self.handle:
                _lib.lib().vae_destroy(self.handle)
                self.handle = None
        except Exception:
            pass


class _VAEForward(torch.autograd.Function):
    """forward (models.py:185-188) + backward (train.py:650) through the C ABI."""

    @staticmethod
    def forward(ctx, x, anchor, model, eps):
        ctx.set_materialize_grads(False)
        ctx.model_ref = weakref.ref(model)      # no model -> _last -> grad_fn -> model reference cycle
        out = model._run_forward(x, eps, train=model.training)
        ctx.generation = model._fwd_count       # the activations saved for backward live in the context, not in the graph
        handle = torch.zeros((), device=x.device, dtype=torch.float32)
        return (*out, handle)

    @staticmethod
    def backward(ctx, g_xhat, g_mu, g_lv, g_z, g_pre, g_handle):
        model = ctx.model_ref()
        if model is None:
            raise RuntimeError("the VanillaVAE of this graph no longer exists")
        if model._fwd_count != ctx.generation:
            raise RuntimeError("backward through a forward that is no longer the model's last one: the context keeps the "
                               "activations of one forward only (run backward before the next forward, or one model per graph)")
        model._run_backward(g_xhat, g_mu, g_lv, g_z, g_pre, g_handle)
        return None, None, None, None


class _FusedELBO(torch.autograd.Function):
    """VanillaVAE.loss (models.py:190-225) of the model's last forward."""

    @staticmethod
    def forward(ctx, handle, model, kld_weight):
        out3 = torch.empty(3, device=handle.device, dtype=torch.float32)
        with torch.cuda.device(handle.device):
            _lib.check(_lib.lib().vae_loss(model._ctx.handle, float(kld_weight), out3.data_ptr(), _stream_ptr(handle.device)), "vae_loss")
        model._bwd_kld_weight = float(kld_weight)
        ctx.mark_non_differentiable(out3)
        return out3[0].clone(), out3

    @staticmethod
    def backward(ctx, g_loss, _g_out3):
        return g_loss, None, None


class _GenericELBO(torch.autograd.Function):
    """VanillaVAE.loss on tensors that are not the model's own last forward."""

    @staticmethod
    def forward(ctx, xhat, target, mu, lv, kld_weight):
        xhat, target, mu, lv = (t.contiguous().float() for t in (xhat, target, mu, lv))
        out3 = torch.empty(3, device=xhat.device, dtype=torch.float32)
        gx, gm, gl = torch.empty_like(xhat), torch.empty_like(mu), torch.empty_like(lv)
        with torch.cuda.device(xhat.device):
            _lib.check(_lib.lib().vae_elbo_generic(xhat.data_ptr(), target.data_ptr(), mu.data_ptr(), lv.data_ptr(),
                                                  xhat.numel(), mu.shape[0], mu.shape[1], float(kld_weight),
                                                  out3.data_ptr(), gx.data_ptr(), gm.data_ptr(), gl.data_ptr(),
                                                  _stream_ptr(xhat.device)), "vae_elbo_generic")
        ctx.save_for_backward(gx, gm, gl)
        ctx.mark_non_differentiable(out3)
        return out3[0].clone(), out3

    @staticmethod
    def backward(ctx, g_loss, _g_out3):
        gx, gm, gl = ctx.saved_tensors
        return gx * g_loss, None, gm * g_loss, gl * g_loss, None


class VanillaVAE(nn.Module):
    name = "VanillaVAE"

    def __init__(
        self,
        in_channels: int,
        embed_dim: int,
        input_dim: int,
        hidden_dims: list[int] = None,
        kld_weight: float = 1.0,
        verbose: bool = False,
        *,
        generalised: bool | None = None,
        compute_dtype: str = "bf16",
        max_batch: int | None = None,
    ):
        super().__init__()
        if in_channels != 1:
            raise NotImplementedError("the MI355X VAE step is built for 1-channel pianoroll input (in_channels=1)")
        self.latent_dim = embed_dim
        self.input_dim = input_dim
        self.in_channels = in_channels
        self.verbose = verbose
        self.kld_weight = kld_weight
        if hidden_dims is None:
            hidden_dims = [32, 64, 128, 256]
        if list(hidden_dims) != [32, 64, 128, 256]:
            raise NotImplementedError("hidden_dims must be the reference default [32, 64, 128, 256]")
        self.hidden_dims = hidden_dims
        if generalised is None:
            generalised = input_dim != 32
        self.generalised = bool(generalised)
        self.img_size = int(input_dim) if self.generalised else 32
        if compute_dtype not in _DTYPES:
            raise ValueError(f"compute_dtype must be one of {sorted(_DTYPES)}")
        self.compute_dtype = compute_dtype
        s = self.img_size // 16 if self.generalised else 2
        self.last_conv_size = s * s  # models.py:33 hard-wires 4
        self.flattened_size = self.last_conv_size * hidden_dims[-1]
        self._offs, self._sizes, self._total = _lib.param_layout(self.img_size, embed_dim, self.generalised)
        self._bn_offs, self._bn_ch, self._bn_total = _lib.bn_layout()

        # parameter containers, built and initialised in the reference's order (models.py:41-83)
        modules = []
        cin = in_channels
        for h_dim in hidden_dims:
            modules.append(nn.Sequential(nn.Conv2d(cin, h_dim, kernel_size=3, stride=2, padding=1),
                                         nn.BatchNorm2d(h_dim), nn.LeakyReLU()))
            cin = h_dim
        self.encoder = nn.Sequential(*modules)
        self._init_weights(self.encoder, "encoder")
        self.fc_mu = nn.Linear(self.flattened_size, self.latent_dim)
        self.fc_var = nn.Linear(self.flattened_size, self.latent_dim)
        self.decoder_input = nn.Linear(self.latent_dim, self.flattened_size)
        hidden_dims.reverse()  # models.py:60 mutates the caller's list; kept
        modules = []
        for i in range(len(hidden_dims) - 1):
            modules.append(nn.Sequential(
                nn.ConvTranspose2d(hidden_dims[i], hidden_dims[i + 1], kernel_size=3, stride=2, padding=1, output_padding=1),
                nn.BatchNorm2d(hidden_dims[i + 1]), nn.LeakyReLU()))
        self.decoder = nn.Sequential(*modules)
        self._init_weights(self.decoder, "decoder")
        self.final_layer = nn.Sequential(
            nn.ConvTranspose2d(hidden_dims[-1], hidden_dims[-1], kernel_size=3, stride=2, padding=1, output_padding=1),
            nn.BatchNorm2d(hidden_dims[-1]), nn.LeakyReLU(),
            nn.Conv2d(hidden_dims[-1], 1, kernel_size=3, stride=1, padding=1), nn.Sigmoid())
        self._init_weights(self.final_layer, "final layer")

        self._flat = self._gflat = self._gnew = self._bnflat = self._nbt = None
        self._ctx = None
        self._max_batch = max_batch
        self._next_eps = None
        self._last = None
        self._bwd_kld_weight = float(kld_weight)
        self.materialize_pre_latents = True
        self.eps_seed = 0
        self._fwd_count = 0

    # -- reference helpers --------------------------------------------------
    def _init_weights(self, module: nn.Sequential, name: str):
        """models.py:227-236: xavier for nn.Linear/nn.Conv2d inside `module`, BN weight 1 / bias 0.
        nn.ConvTranspose2d is not an nn.Conv2d subclass and keeps its default init."""
        for m in module.modules():
            if isinstance(m, nn.Linear) or isinstance(m, nn.Conv2d):
                nn.init.xavier_uniform_(m.weight)
                if m.bias is not None:
                    nn.init.constant_(m.bias, 0)
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)
        if self.verbose:
            print(f"{name} layer weight initialization complete")

    def _compute_conv_output_size(self, dim: int, num_layers: int, kernel: int = 3, stride: int = 2, padding: int = 1):
        for _ in range(num_layers):
            dim = (dim - kernel + stride * padding) // stride + 1
        return dim

    # -- flat storage -------------------------------------------------------
    def _named_param_list(self):
        sd = dict(self.named_parameters())
        return [sd[n] for n in _lib.PARAM_NAMES]

    def _param_grad_views(self):
        """(parameters, their slices of the flat gradient buffer), cached: walking named_parameters() and re-slicing on
        every step cost ~0.1 ms of host time, more than the launches of a 32x32 step.  Rebuilt with the flat buffers."""
        cache = self.__dict__.get("_gview_cache")
        if cache is None or cache[0] is not self._gflat:
            params = self._named_param_list()
            views = [self._gflat[self._offs[i]:self._offs[i] + self._sizes[i]].view(p.shape) for i, p in enumerate(params)]
            cache = (self._gflat, params, views)
            self.__dict__["_gview_cache"] = cache
        return cache[1], cache[2]

    def _bn_modules(self):
        return [self.get_submodule(n) for n in _lib.BN_NAMES]

    def _apply(self, fn, *args, **kwargs):
        out = super()._apply(fn, *args, **kwargs)
        self._rebuild_flat()
        return out

    def _rebuild_flat(self):
        params = self._named_param_list()
        dev = params[0].device
        if dev.type != "cuda":
            self._flat = None
            return
        if any(p.dtype != torch.float32 for p in params):
            raise TypeError("VanillaVAE parameters must stay float32 (master weights); pick compute_dtype instead")
        flat = torch.zeros(self._total, device=dev, dtype=torch.float32)
        for i, p in enumerate(params):
            view = flat[self._offs[i]:self._offs[i] + self._sizes[i]].view(p.shape)
            view.copy_(p.data)
            p.data = view
            p._vae_owner = weakref.ref(self)
            p._vae_index = i
            p.grad = None
        bnflat = torch.zeros(self._bn_total, device=dev, dtype=torch.float32)
        nbt = torch.zeros(_lib.NUM_BN, device=dev, dtype=torch.int64)
        for i, bn in enumerate(self._bn_modules()):
            c, o = self._bn_ch[i], self._bn_offs[i]
            bnflat[o:o + c].copy_(bn.running_mean)
            bnflat[o + c:o + 2 * c].copy_(bn.running_var)
            nbt[i].copy_(bn.num_batches_tracked)
            bn.running_mean = bnflat[o:o + c]
            bn.running_var = bnflat[o + c:o + 2 * c]
            bn.num_batches_tracked = nbt[i]
        self._flat, self._bnflat, self._nbt = flat, bnflat, nbt
        self._gflat = torch.zeros_like(flat)
        self._gnew = torch.zeros_like(flat)
        self._ctx = None
        self._last = None

    def flat_parameters(self) -> Tensor:
        self._require_device()
        return self._flat

    def flat_grads(self) -> Tensor:
        self._require_device()
        return self._gflat

    def group_range(self, prefix: str) -> tuple[int, int]:
        """(offset, length) in the flat buffer of all parameters whose name starts with `prefix`."""
        idx = [i for i, n in enumerate(_lib.PARAM_NAMES) if n.startswith(prefix + ".")]
        if not idx or idx != list(range(idx[0], idx[-1] + 1)):
            raise ValueError(f"{prefix!r} is not a contiguous parameter range")
        start = self._offs[idx[0]]
        return start, self._offs[idx[-1]] + self._sizes[idx[-1]] - start

    def comm_stream(self) -> "torch.cuda.Stream":
        """The context's communication stream, ordered after the work enqueued so far on the current stream; joined
        back at the end of the backward's second half (include/vae_step.h: vae_comm_stream)."""
        import ctypes as C
        out = C.c_void_p()
        with self._device_guard():
            _lib.check(_lib.lib().vae_comm_stream(self._ctx.handle, self._stream(), C.byref(out)), "vae_comm_stream")
        return torch.cuda.ExternalStream(out.value, device=self._flat.device)

    def _require_device(self):
        if self._flat is None:
            raise RuntimeError("VanillaVAE (MI355X) runs only on a HIP device: call model.to('cuda') first; "
                               "there is no CPU path.")

    def _device_guard(self):
        """Every library call runs with the MODEL's device current: vae_create allocates its workspace on the current
        HIP device and kernels launch on the stream passed in, which must belong to the device that owns the tensors."""
        dev = self._flat.device
        if torch.cuda.current_device() == dev.index:
            return _NULL_GUARD          # (the usual case: entering torch.cuda.device costs two HIP calls per library call)
        return torch.cuda.device(dev)

    def _stream(self):
        return _stream_ptr(self._flat.device)

    def _context(self, batch: int) -> _Context:
        need = max(batch, self._max_batch or 0)
        if self._ctx is None or self._ctx.key[2] < batch:
            self._ctx = None
            with self._device_guard():
                self._ctx = _Context(self.img_size, self.latent_dim, need, _DTYPES[self.compute_dtype], self.generalised)
                if getattr(self, "_want_lib_comm", False):
                    self._init_library_comm()
            self._max_batch = need
        return self._ctx

    # -- data parallel: the step library's own RCCL communicator (include/vae_step.h: vae_comm_*) ----------------------
    def _init_library_comm(self):
        """COLLECTIVE over torch.distributed's default group: rank 0 draws the RCCL unique id, the process group carries it
        to the other ranks, every rank creates its communicator.  On failure the model falls back to torch.distributed's
        collectives (and says so once)."""
        import ctypes as C
        import warnings
        import torch.distributed as dist
        L = _lib.lib()
        rank, world = dist.get_rank(), dist.get_world_size()
        buf = C.create_string_buffer(_lib.COMM_ID_BYTES)
        ok = True
        if rank == 0:
            ok = L.vae_comm_unique_id(buf) == 0
        box = [bytes(buf.raw) if ok else None]
        dist.broadcast_object_list(box, src=0)
        if box[0] is None or L.vae_comm_init(self._ctx.handle, rank, world, C.create_string_buffer(box[0], _lib.COMM_ID_BYTES)) != 0:
            warnings.warn("torch_vae_amd: RCCL communicator of the step library unavailable ("
                          + L.vae_last_error().decode() + "); gradients go through torch.distributed instead")
            self._want_lib_comm = False

    def library_comm_world(self) -> int:
        """World size of the context's own RCCL communicator (0: none, gradients go through torch.distributed)."""
        if self._ctx is None or not self._ctx.handle:
            return 0
        return int(_lib.lib().vae_comm_world(self._ctx.handle))

    def allreduce_ranges(self, prefixes, on_comm_stream: bool = False, average: bool = True):
        """Mean (or sum) all-reduce of the flat-gradient ranges of the named parameter groups through the library's
        communicator, as ONE RCCL group, on the current stream or on the context's communication stream (which is ordered
        after everything enqueued so far and joined back by the second half of the backward)."""
        import ctypes as C
        L = _lib.lib()
        rngs = [self.group_range(p) for p in prefixes]
        offs = (C.c_int64 * len(rngs))(*[r[0] for r in rngs])
        sizes = (C.c_int64 * len(rngs))(*[r[1] for r in rngs])
        with self._device_guard():
            st = self._stream()
            if on_comm_stream:
                out = C.c_void_p()
                _lib.check(L.vae_comm_stream(self._ctx.handle, st, C.byref(out)), "vae_comm_stream")
                st = out.value
            _lib.check(L.vae_allreduce_grads(self._ctx.handle, self._gflat.data_ptr(), len(rngs), offs, sizes, int(average), st),
                       "vae_allreduce_grads")

    # -- kernels ------------------------------------------------------------
    def set_next_eps(self, eps: Tensor | None):
        """Supply the N(0,1) draw models.py:182 takes from torch.randn_like (parity runs)."""
        self._next_eps = eps

    def _check_input(self, x: Tensor):
        self._require_device()
        if x.dim() != 4 or x.shape[1] != 1 or x.shape[2] != self.img_size or x.shape[3] != self.img_size:
            # the reference fails inside ATen with a shape RuntimeError; same exception type
            raise RuntimeError(f"expected input [B,1,{self.img_size},{self.img_size}], got {tuple(x.shape)}")
        if x.device != self._flat.device:
            raise RuntimeError("input and model are on different devices")

    def _run_forward(self, x: Tensor, eps: Tensor | None, train: bool, want_pre: bool | None = None, defer_output: bool = False):
        self._check_input(x)
        x = x.detach().contiguous().float()
        B, L, dev = x.shape[0], self.latent_dim, x.device
        ctx = self._context(B)
        xhat = torch.empty_like(x)
        mu = torch.empty(B, L, device=dev)
        lv = torch.empty(B, L, device=dev)
        z = torch.empty(B, L, device=dev)
        if eps is not None:
            eps = eps.detach().to(dev, torch.float32).contiguous()
            if eps.shape != (B, L):
                raise RuntimeError(f"eps must be [{B},{L}]")
        self._fwd_count += 1
        # seed of the device-side reparameterisation noise (used when eps is None): distinct per step AND per rank, so
        # data-parallel replicas draw independent noise like the reference's per-process torch generator (models.py:182)
        seed = (int(self.eps_seed) + self._fwd_count + _rank() * 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF
        with self._device_guard():
            _lib.check(_lib.lib().vae_forward(
                ctx.handle, x.data_ptr(), B, self._flat.data_ptr(), self._bnflat.data_ptr(), self._nbt.data_ptr(),
                _lib.ptr(eps), seed, (2 if (train and defer_output) else int(train)), xhat.data_ptr(), mu.data_ptr(),
                lv.data_ptr(), z.data_ptr(), self._stream()), "vae_forward")
            if want_pre is None:
                want_pre = self.materialize_pre_latents
            if want_pre:
                pre = torch.empty(B, self.flattened_size, device=dev)
                _lib.check(_lib.lib().vae_pre_latents(ctx.handle, pre.data_ptr(), self._stream()), "vae_pre_latents")
            else:
                pre = torch.empty(B, 0, device=dev)
        self._last = dict(x=x, xhat=xhat, mu=mu, lv=lv, z=z, train=train)
        return xhat, mu, lv, z, pre

    def _run_backward(self, g_xhat, g_mu, g_lv, g_z, g_pre, g_handle, into_gflat: bool = False):
        last = self._last
        if last is None or not last["train"]:
            raise RuntimeError("backward needs a train-mode forward of this model")
        c = lambda t: None if t is None else t.contiguous().float()  # noqa: E731
        g_xhat, g_mu, g_lv, g_z, g_pre, g_handle = map(c, (g_xhat, g_mu, g_lv, g_z, g_pre, g_handle))
        target = self._gflat if into_gflat else self._gnew
        with self._device_guard():
            _lib.check(_lib.lib().vae_backward(
                self._ctx.handle, last["x"].data_ptr(), self._flat.data_ptr(), target.data_ptr(), _lib.ptr(g_xhat),
                _lib.ptr(g_handle), _lib.ptr(g_mu), _lib.ptr(g_lv), _lib.ptr(g_z), _lib.ptr(g_pre),
                float(self._bwd_kld_weight), int(g_handle is not None), self._stream()), "vae_backward")
        if not into_gflat:
            self._commit_grads()

    def _commit_grads(self):
        """param.grad semantics of autograd: assign where grad is None, accumulate otherwise."""
        params = self._named_param_list()
        runs, cur = [], None
        for i, p in enumerate(params):
            if not p.requires_grad:
                kind = "skip"
            elif p.grad is None:
                kind = "assign"
            elif p.grad.data_ptr() == self._gflat.data_ptr() + 4 * self._offs[i]:
                kind = "add"
            else:
                kind = "foreign"
            if kind == "foreign":
                p.grad.add_(self._gnew[self._offs[i]:self._offs[i] + self._sizes[i]].view(p.shape))
                cur = None
                continue
            if cur is not None and cur[0] == kind:
                cur[2] = self._offs[i] + self._sizes[i]
            else:
                cur = [kind, self._offs[i], self._offs[i] + self._sizes[i]]
                runs.append(cur)
        for kind, a, b in runs:
            if kind == "assign":
                self._gflat[a:b].copy_(self._gnew[a:b])
            elif kind == "add":
                self._gflat[a:b].add_(self._gnew[a:b])
        for i, p in enumerate(params):
            if p.requires_grad and p.grad is None:
                p.grad = self._gflat[self._offs[i]:self._offs[i] + self._sizes[i]].view(p.shape)

    def bind_flat_grads(self):
        """Point every param.grad at its slice of the flat gradient buffer (fused step path)."""
        params, views = self._param_grad_views()
        for p, g in zip(params, views):
            if p.grad is not g and p.requires_grad:
                if p.grad is None or p.grad.data_ptr() != g.data_ptr():
                    p.grad = g

    # -- reference API ------------------------------------------------------
    def encode(self, x: Tensor) -> EncoderOutput:
        """models.py:107-145 (runs the whole fused forward; the decoder half is discarded)."""
        out = self.forward(x)
        return out["encoded"]

    def decode(self, z: Tensor) -> Tensor:
        """models.py:147-175: latent [B, latent_dim] -> reconstruction [B,1,H,W].  Inference helper (SURVEY.md 8f,
        N4): runs the decoder kernels only; not differentiable (training differentiates through forward())."""
        self._require_device()
        z = z.detach().to(self._flat.device, torch.float32).contiguous()
        if z.dim() != 2 or z.shape[1] != self.latent_dim:
            raise RuntimeError(f"expected z [B,{self.latent_dim}], got {tuple(z.shape)}")
        B = z.shape[0]
        ctx = self._context(B)
        xhat = torch.empty(B, 1, self.img_size, self.img_size, device=z.device, dtype=torch.float32)
        with self._device_guard():
            _lib.check(_lib.lib().vae_decode(ctx.handle, z.data_ptr(), B, self._flat.data_ptr(), self._bnflat.data_ptr(),
                                            self._nbt.data_ptr(), int(self.training), xhat.data_ptr(), self._stream()), "vae_decode")
        self._last = None
        return xhat

    def reparameterize(self, mu: Tensor, log_var: Tensor) -> Tensor:
        """models.py:177-183 (plumbing-level torch ops; the hot path fuses this into the latent kernel)."""
        std = torch.exp(0.5 * log_var)
        return torch.randn_like(std) * std + mu

    def forward(self, x: Tensor) -> ModelOutput:
        self._require_device()
        eps, self._next_eps = self._next_eps, None
        if eps is None:
            # torch.randn_like(std) of models.py:182: drawn from torch's device generator
            eps = torch.randn(x.shape[0], self.latent_dim, device=x.device, dtype=torch.float32)
        if torch.is_grad_enabled() and self.training:
            anchor = self._anchor_tensor(x.device)
            xhat, mu, lv, z, pre, handle = _VAEForward.apply(x, anchor, self, eps)
        else:
            xhat, mu, lv, z, pre = self._run_forward(x, eps, train=self.training)
            handle = None
        self._last.update(handle=handle, out_ref=xhat, mu=mu, lv=lv, z=z)
        encoding = EncoderOutput(mu=mu, log_var=lv, pre_latents=pre)
        return ModelOutput(output=xhat, input=x, encoded=encoding, latents=z)

    def _anchor_tensor(self, dev):
        a = getattr(self, "_anchor", None)
        if a is None or a.device != dev:
            a = torch.zeros((), device=dev, requires_grad=True)
            self._anchor = a
        return a

    def loss(self, output: ModelOutput):
        """models.py:190-225."""
        last = self._last
        own = (last is not None and output["output"] is last.get("out_ref") and last.get("handle") is not None
               and output["encoded"]["mu"] is last["mu"] and (output["input"] is last["x"] or
                                                            output["input"].data_ptr() == last["x"].data_ptr()))
        if own:
            loss, out3 = _FusedELBO.apply(last["handle"], self, self.kld_weight)
        elif last is not None and output["output"] is last.get("out_ref") and last.get("handle") is None:
            out3 = torch.empty(3, device=last["xhat"].device)
            with self._device_guard():
                _lib.check(_lib.lib().vae_loss(self._ctx.handle, float(self.kld_weight), out3.data_ptr(), self._stream()), "vae_loss")
            loss = out3[0].clone()
        else:
            loss, out3 = _GenericELBO.apply(output["output"], output["input"], output["encoded"]["mu"],
                                            output["encoded"]["log_var"], self.kld_weight)
        return LossOutput(loss=loss, reconstruction_loss=out3[1].detach(), kld_loss=out3[2].detach())

    def sample(self, num_samples: int, current_device: int, **kwargs) -> Tensor:
        """models.py:250-263: z ~ N(0, I) on the host generator, moved to the device, decoded."""
        z = torch.randn(num_samples, self.latent_dim)
        z = z.to(current_device)
        return self.decode(z)

    def generate(self, x: Tensor, **kwargs) -> Tensor:
        """models.py:265-272."""
        return self.forward(x)["output"]

    # -- fused step (no autograd): forward, ELBO, backward in one call chain --
    def fused_forward_backward(self, x: Tensor, eps: Tensor | None = None, use_device_eps: bool = True,
                               on_decoder_grads=None):
        """forward -> loss -> backward (train.py:634-650) with gradients written straight into the
        flat gradient buffer.  Returns a 3-element device tensor {loss, reconstruction, kld_loss}.
        ``on_decoder_grads`` (data parallel): called between the two halves of the backward, when every
        decoder / final_layer gradient is complete in stream order - the caller starts that bucket's
        all-reduce there so it overlaps the encoder half."""
        self._require_device()
        if eps is None and not use_device_eps:
            eps = torch.randn(x.shape[0], self.latent_dim, device=x.device, dtype=torch.float32)
        # train = 2: the library may leave the output conv / sigmoid / BCE to the backward (one kernel for the layer's
        # forward and backward): xhat and the ELBO scalars are then written by the backward call below
        xhat, mu, lv, z, _ = self._run_forward(x, eps, train=True, want_pre=False, defer_output=True)
        out3 = torch.empty(3, device=x.device, dtype=torch.float32)
        # the ELBO scalars are only read after the step: finalised beside the backward (joined by it), not in front of it
        with self._device_guard():
            _lib.check(_lib.lib().vae_loss_deferred(self._ctx.handle, float(self.kld_weight), out3.data_ptr(), self._stream()), "vae_loss_deferred")
        self._bwd_kld_weight = float(self.kld_weight)
        last = self._last
        for part in ((0,) if on_decoder_grads is None else (1, 2)):
            with self._device_guard():
                _lib.check(_lib.lib().vae_backward_part(
                    self._ctx.handle, last["x"].data_ptr(), self._flat.data_ptr(), self._gflat.data_ptr(), 0, 0, 0, 0, 0, 0,
                    float(self.kld_weight), 1, part, self._stream()), "vae_backward")
            if part == 1:
                self.bind_flat_grads()
                on_decoder_grads()
        self.bind_flat_grads()
        return out3, xhat


    # -- the whole training step as ONE library call ------------------------------------------------------------
    def fused_train_step(self, optimizer, x: Tensor, eps: Tensor | None = None, use_device_eps: bool = True, exchange: int = 0):
        """forward -> ELBO -> backward -> [gradient exchange] -> AdamW (train.py:634-656) enqueued by ONE call into the library
        (include/vae_step.h: vae_train_step_fused) instead of five, with no per-step tensor allocation: the outputs live in
        buffers owned by the model for the batch size, valid until the next fused step of this model.  ``exchange``: 0 none,
        1 one RCCL group between backward and AdamW, 2 bucketed (decoder bucket under the encoder backward, each group's AdamW
        behind its own bucket) - both need the library communicator.  Returns (out3, xhat)."""
        import ctypes as C
        self._check_input(x)
        if not x.is_contiguous() or x.dtype != torch.float32:
            x = x.detach().contiguous().float()
        B, L, dev = x.shape[0], self.latent_dim, x.device
        ctx = self._context(B)
        if eps is None and not use_device_eps:
            eps = torch.randn(B, L, device=dev, dtype=torch.float32)
        if eps is not None:
            eps = eps.detach().to(dev, torch.float32).contiguous()
            if eps.shape != (B, L):
                raise RuntimeError(f"eps must be [{B},{L}]")
        bufs = self.__dict__.get("_step_bufs")
        if bufs is None or bufs[0] != (B, dev):
            bufs = ((B, dev), torch.empty(B, 1, self.img_size, self.img_size, device=dev), torch.empty(B, L, device=dev),
                    torch.empty(B, L, device=dev), torch.empty(B, L, device=dev), torch.empty(3, device=dev))
            self.__dict__["_step_bufs"] = bufs
        _, xhat, mu, lv, z, out3 = bufs
        self._fwd_count += 1
        seed = (int(self.eps_seed) + self._fwd_count + _rank() * 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF
        n, offs, sizes, lrs, b1s, beta2, adam_eps, wd = optimizer._step_args()
        self._bwd_kld_weight = float(self.kld_weight)
        with self._device_guard():
            _lib.check(_lib.lib().vae_train_step_fused(
                ctx.handle, x.data_ptr(), B, self._flat.data_ptr(), self._gflat.data_ptr(), optimizer._m.data_ptr(),
                optimizer._v.data_ptr(), self._bnflat.data_ptr(), self._nbt.data_ptr(), _lib.ptr(eps), seed, float(self.kld_weight),
                n, offs, sizes, lrs, b1s, beta2, adam_eps, wd, float(optimizer.grad_scale), optimizer._step + 1, int(exchange),
                xhat.data_ptr(), mu.data_ptr(), lv.data_ptr(), z.data_ptr(), out3.data_ptr(), self._stream()), "vae_train_step_fused")
        optimizer._stepped()
        self._last = dict(x=x, xhat=xhat, mu=mu, lv=lv, z=z, train=True)
        params, views = self._param_grad_views()
        if params[0].grad is not views[0] or params[-1].grad is not views[-1]:   # (bound once: the fused path never unbinds them)
            self.bind_flat_grads()
        return out3, xhat


def count_flops_per_sample(img_size: int, latent_dim: int, generalised: bool = True) -> float:
    """Algorithmic FLOPs of one training step per sample: 3 x 2 x forward MACs."""
    h = img_size
    s = h // 16 if generalised else 2
    macs = 9 * 1 * 32 * (h // 2) ** 2
    for ci, co, ho in ((32, 64, h // 4), (64, 128, h // 8), (128, 256, h // 16)):
        macs += 9 * ci * co * ho * ho
    f = 256 * s * s
    macs += 2 * f * latent_dim + f * latent_dim
    for ci, co, hi in ((256, 128, s), (128, 64, 2 * s), (64, 32, 4 * s), (32, 32, 8 * s)):
        macs += 9 * ci * co * hi * hi
    macs += 9 * 32 * (16 * s) ** 2
    return 6.0 * macs


def algorithmic_bytes_per_step(img_size: int, latent_dim: int, batch: int, elem_bytes: int, generalised: bool = True) -> float:
    """B*(2*H^2*e + 5*A*e) + P*(2e + 32)."""
    h = img_size
    s = h // 16 if generalised else 2
    f = 256 * s * s
    a = 32 * (h // 2) ** 2 + 64 * (h // 4) ** 2 + 128 * (h // 8) ** 2 + 256 * (h // 16) ** 2
    a += 3 * latent_dim + f
    a += 128 * (2 * s) ** 2 + 64 * (4 * s) ** 2 + 32 * (8 * s) ** 2 + 32 * (16 * s) ** 2 + (16 * s) ** 2
    _, sizes, _ = _lib.param_layout(img_size if generalised else 32, latent_dim, generalised)
    p = sum(sizes)
    return batch * (2 * h * h * elem_bytes + 5 * a * elem_bytes) + p * (2 * elem_bytes + 32)
