"""Diagnostic (GPU box): which internal tensors differ between use_fused_convout = 0 / 1."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from oracle import vae_oracle as vo
from tests.util import make_model, perturbed_params
from torch_vae_amd import _lib
from torch_vae_amd._lib import PARAM_NAMES
H, L, B = 64, 16, 6
p = perturbed_params(L, H, 11, True)
x = torch.from_numpy(vo.synth_pianoroll(B, H, 5)).cuda()
eps = torch.from_numpy(vo.counter_normal(B * L, 5, 5).reshape(B, L)).float().cuda()
C = [32, 64, 128, 256, 128, 64, 32, 32]; HH = [H // 2, H // 4, H // 8, H // 16, H // 8, H // 4, H // 2, H]
for dtype in ("bf16", "f16"):
    outs = []
    for fused in (0, 1):
        m = make_model(H, L, True, dtype, p, kld_weight=2.0)
        _lib.check(_lib.lib().vae_set_option(m._context(B).handle, b"use_fused_convout", fused), "set")
        m.fused_forward_backward(x, eps=eps)
        t = {}
        for i in range(8):
            n = B * C[i] * HH[i] * HH[i]
            for which, nm in ((i, f"y{i}"), (8 + i, f"dz{i}")):
                buf = torch.empty(n, device="cuda")
                _lib.check(_lib.lib().vae_debug_tensor(m._ctx.handle, which, buf.data_ptr(), n, torch.cuda.current_stream().cuda_stream), "dbg")
                t[nm] = buf
        t["grads"] = m.flat_grads().clone()
        outs.append((t, m))
    (t0, m0), (t1, m1) = outs
    for k in t0:
        if k != "grads" and not torch.equal(t0[k], t1[k]):
            d = (t0[k] - t1[k]); print(dtype, k, "differs: n =", int((d != 0).sum()), "max rel", float(d.abs().max() / t0[k].abs().max()))
    for n, o, sz in zip(PARAM_NAMES, m0._offs, m0._sizes):
        a, b = t0["grads"][o:o + sz], t1["grads"][o:o + sz]
        if not torch.equal(a, b):
            idx = (a != b).nonzero().flatten()
            print(dtype, n, "differs at", idx[:5].tolist(), "of", sz, "n", len(idx), a[idx[:3]].tolist(), b[idx[:3]].tolist())
