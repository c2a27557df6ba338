"""Fused AdamW over the model's flat parameter buffer.

Behaves as ``torch.optim.AdamW`` (train.py:228): same ``param_groups`` keys, so
``torch.optim.lr_scheduler.OneCycleLR`` (train.py:233-238) drives ``lr`` and
``betas[0]`` exactly as it does for the reference.  ``step()`` is one kernel launch
for all groups (include/vae_step.h: vae_adamw_step).
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib


class FusedAdamW(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2):
        # (the inert keys are torch.optim.AdamW's: checkpoints written by either optimiser then carry the same param_group
        #  keys and load into the other, reference train.py:320-329)
        defaults = dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, amsgrad=False, maximize=False, foreach=None,
                        capturable=False, differentiable=False, fused=None, decoupled_weight_decay=True)
        super().__init__(params, defaults)
        self._model = None
        self._m = self._v = None
        self._step = 0
        self.grad_scale = 1.0
        if len(self.param_groups) > 2:
            raise ValueError("FusedAdamW supports at most two parameter groups (encoder, decoder)")

    def _bind(self):
        if self._model is not None:
            return
        owners = {getattr(p, "_vae_owner", None) for g in self.param_groups for p in g["params"]}
        if len(owners) != 1 or None in owners:
            raise ValueError("FusedAdamW needs parameters of one torch_vae_amd VanillaVAE that is already on the GPU "
                             "(move the model with .to('cuda') before building the optimiser)")
        self._model = next(iter(owners))()
        model = self._model
        self._ranges = []
        for g in self.param_groups:
            idx = sorted(p._vae_index for p in g["params"])
            if idx != list(range(idx[0], idx[-1] + 1)):
                raise ValueError("each FusedAdamW group must be a contiguous run of model parameters")
            start = model._offs[idx[0]]
            self._ranges.append((start, model._offs[idx[-1]] + model._sizes[idx[-1]] - start))
        flat = model.flat_parameters()
        self._m = torch.zeros_like(flat)
        self._v = torch.zeros_like(flat)
        # one step counter tensor shared by every parameter's state (torch keeps one per parameter and bumps each: 16 host
        # tensor ops per step); state_dict() shows the same values either way
        self._step_t = torch.tensor(0.0)
        for g in self.param_groups:
            for p in g["params"]:
                o, n = model._offs[p._vae_index], model._sizes[p._vae_index]
                self.state[p] = {"step": self._step_t, "exp_avg": self._m[o:o + n].view(p.shape),
                                 "exp_avg_sq": self._v[o:o + n].view(p.shape)}

    def load_state_dict(self, state_dict):
        """torch's loader replaces the state tensors; copy them back into the flat moment buffers."""
        self._bind()
        super().load_state_dict(state_dict)
        model = self._model
        step = 0
        for g in self.param_groups:
            for p in g["params"]:
                st = self.state.get(p, {})
                o, n = model._offs[p._vae_index], model._sizes[p._vae_index]
                for key, flat in (("exp_avg", self._m), ("exp_avg_sq", self._v)):
                    view = flat[o:o + n].view(p.shape)
                    if key in st:
                        view.copy_(st[key])
                    st[key] = view
                step = max(step, int(st.get("step", 0)))
                self.state[p] = st
        self._step = step
        self._step_t = torch.tensor(float(step))
        for g in self.param_groups:
            for p in g["params"]:
                self.state[p]["step"] = self._step_t

    def _step_args(self):
        """(ngroups, offsets, sizes, lrs, beta1s, beta2, eps, weight_decay) of the next update as the C ABI wants them, for
        the one-call fused step (VanillaVAE.fused_train_step): every group is active there, the gradients being the flat
        buffer's.  The ctypes arrays are cached; only the two scheduler-driven values per group are refreshed."""
        self._bind()
        cache = self.__dict__.get("_arg_cache")
        n = len(self.param_groups)
        if cache is None:
            cache = ((C.c_int64 * n)(*[r[0] for r in self._ranges]), (C.c_int64 * n)(*[r[1] for r in self._ranges]),
                     (C.c_double * n)(), (C.c_double * n)())
            self.__dict__["_arg_cache"] = cache
        offs, sizes, lrs, b1s = cache
        g0 = self.param_groups[0]
        for i, g in enumerate(self.param_groups):
            lrs[i] = g["lr"]; b1s[i] = g["betas"][0]
            if (g["betas"][1], g["eps"], g["weight_decay"]) != (g0["betas"][1], g0["eps"], g0["weight_decay"]):
                raise NotImplementedError("groups must share beta2, eps and weight_decay")
        return n, offs, sizes, lrs, b1s, float(g0["betas"][1]), float(g0["eps"]), float(g0["weight_decay"])

    def _stepped(self):
        """Bookkeeping of an update the library performed inside vae_train_step_fused."""
        self._step += 1
        self._step_t += 1
        self._opt_called = True   # (torch's schedulers check that the optimiser stepped before them)

    @torch.no_grad()
    def step(self, closure=None):
        loss = closure() if closure is not None else None
        self._bind()
        model = self._model
        gflat = model.flat_grads()
        views = model._param_grad_views()[1]
        active = []
        for g, rng in zip(self.param_groups, self._ranges):
            grads = [p.grad for p in g["params"]]
            if all(gr is None for gr in grads):
                continue  # torch skips parameters without gradients
            for p, gr in zip(g["params"], grads):
                if gr is views[p._vae_index]:
                    continue              # the fused path's own view of the flat gradient buffer
                o, n = model._offs[p._vae_index], model._sizes[p._vae_index]
                if gr is None:
                    raise NotImplementedError("a FusedAdamW group with only some gradients set")
                if gr.data_ptr() != gflat.data_ptr() + 4 * o:
                    gflat[o:o + n].view(p.shape).copy_(gr)  # foreign gradient tensor: stage it
            active.append((g, rng))
        if not active:
            return loss
        self._step += 1
        n = len(active)
        offs = (C.c_int64 * n)(*[r[0] for _, r in active])
        sizes = (C.c_int64 * n)(*[r[1] for _, r in active])
        lrs = (C.c_double * n)(*[float(g["lr"]) for g, _ in active])
        b1s = (C.c_double * n)(*[float(g["betas"][0]) for g, _ in active])
        g0 = active[0][0]
        for g, _ in active:
            if (g["betas"][1], g["eps"], g["weight_decay"]) != (g0["betas"][1], g0["eps"], g0["weight_decay"]):
                raise NotImplementedError("groups must share beta2, eps and weight_decay")
        dev = gflat.device
        with model._device_guard():    # launch on the model's device, whatever the caller's current device is
            _lib.check(_lib.lib().vae_adamw_step(
                model.flat_parameters().data_ptr(), gflat.data_ptr(), self._m.data_ptr(), self._v.data_ptr(), n, offs, sizes,
                lrs, b1s, float(g0["betas"][1]), float(g0["eps"]), float(g0["weight_decay"]), float(self.grad_scale),
                self._step, torch.cuda.current_stream(dev).cuda_stream), "vae_adamw_step")
        self._step_t += 1
        return loss
