"""Diagnostic (GPU box): fused gradient kernel variants against the separate kernels, per storage type and size."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from oracle import vae_oracle as vo
from tests.util import perturbed_params, make_model, flat_grad_dict, rel_l2
from torch_vae_amd import _lib
for (H, L, B, dtype) in [(64, 16, 5, "bf16"), (64, 16, 5, "f16"), (128, 16, 3, "bf16"), (128, 16, 3, "f16")]:
    p = perturbed_params(L, H, 14, True)
    x = torch.from_numpy(vo.synth_pianoroll(B, H, 19)).cuda()
    eps = torch.from_numpy(vo.counter_normal(B * L, 19, 5).reshape(B, L)).float().cuda()
    res = []
    for use, recomp in ((0, 0), (1, 0), (1, 1)):
        m = make_model(H, L, True, dtype, p); m._context(B)
        _lib.lib().vae_set_option(m._ctx.handle, b"use_fused_wgrad", use); _lib.lib().vae_set_option(m._ctx.handle, b"use_recomp_dz", recomp)
        m.fused_forward_backward(x, eps=eps)
        n = B * 32 * (H // 2) ** 2
        t = torch.empty(n, device="cuda")
        _lib.check(_lib.lib().vae_debug_tensor(m._ctx.handle, 14, t.data_ptr(), n, torch.cuda.current_stream().cuda_stream), "dbg")
        res.append((t.cpu().numpy().reshape(B, 32, H // 2, H // 2), flat_grad_dict(m)))
    for k in (1, 2):
        d = res[k][0] != res[0][0]
        where = np.argwhere(d)
        print(H, dtype, "variant", k, "dz6 mismatches", int(d.sum()), "of", d.size,
              "rows", sorted(set(where[:, 2].tolist()))[:12], "cols", sorted(set(where[:, 3].tolist()))[:12], "imgs", sorted(set(where[:, 0].tolist())),
              "fl0.w rel", rel_l2(res[k][1]["final_layer.0.weight"], res[0][1]["final_layer.0.weight"]))
