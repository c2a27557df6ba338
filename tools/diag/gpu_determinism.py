"""Diagnostic (GPU box): run-to-run bit-identity of the fused step's gradients, per parameter tensor."""
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
for dtype in ("bf16", "f16"):
    for fused in (0, 1):
        ref = None
        for rep in range(6):
            m = make_model(H, L, True, dtype, p, kld_weight=2.0)
            _lib.check(_lib.lib().vae_set_option(m._context(B).handle, b"use_fused_convout", fused), "set")
            for k, v in (dict(kv.split("=") for kv in sys.argv[1:])).items():
                _lib.check(_lib.lib().vae_set_option(m._ctx.handle, k.encode(), int(v)), "set")
            m.fused_forward_backward(x, eps=eps)
            g = m.flat_grads().clone()
            if ref is None:
                ref = g
            else:
                bad = [n for n, o, sz in zip(PARAM_NAMES, m._offs, m._sizes) if not torch.equal(ref[o:o + sz], g[o:o + sz])]
                print(dtype, "fused_convout", fused, "rep", rep, "differing:", bad[:6], len(bad), flush=True)
