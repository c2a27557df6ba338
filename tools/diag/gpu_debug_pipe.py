"""Diagnostic (GPU box): pipelined vs one-tile-per-workgroup conv kernels, tensor by tensor."""
import sys, os
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from oracle import vae_oracle as vo
from torch_vae_amd import _lib
from util import make_model, perturbed_params
import json
H, L, B, gen, dtype = (json.loads(sys.argv[1]) if len(sys.argv) > 1 else [64, 16, 5, True, "bf16"])
OPTS = json.loads(sys.argv[2]) if len(sys.argv) > 2 else {}
p = perturbed_params(L, H, 8, gen)
x = torch.from_numpy(vo.synth_pianoroll(B, H, 12)).cuda()
eps = torch.from_numpy(vo.counter_normal(B * L, 12, 5).reshape(B, L)).float().cuda()
res = []
for use in (0, 1):
    model = make_model(H, L, gen, dtype, p)
    model._context(B)
    assert _lib.lib().vae_set_option(model._ctx.handle, b"use_pipelined", use) == 0
    for k, v in OPTS.items():
        assert _lib.lib().vae_set_option(model._ctx.handle, k.encode(), v) == 0
    out3, xhat = model.fused_forward_backward(x, eps=eps)
    torch.cuda.synchronize()
    t = {}
    C = [32, 64, 128, 256, 128, 64, 32, 32]; S = [H // 2, H // 4, H // 8, H // 16, H // 8, H // 4, H // 2, H]
    for i in range(16):
        n = B * C[i & 7] * S[i & 7] ** 2
        out = torch.empty(n, device="cuda")
        _lib.check(_lib.lib().vae_debug_tensor(model._ctx.handle, i, out.data_ptr(), n, torch.cuda.current_stream().cuda_stream), "dbg")
        t[i] = out.cpu().numpy()
    t["xhat"] = xhat.cpu().numpy().ravel()
    bn = model._bnflat.cpu().numpy(); o = 0
    for li, Cc in enumerate(C):
        t[f"rm{li}"] = bn[o:o + Cc].copy(); t[f"rv{li}"] = bn[o + Cc:o + 2 * Cc].copy(); o += 2 * Cc
    n = B * 256 * (H // 16) ** 2
    out = torch.empty(n, device="cuda")
    _lib.check(_lib.lib().vae_debug_tensor(model._ctx.handle, 17, out.data_ptr(), n, torch.cuda.current_stream().cuda_stream), "dbg")
    t["dd0"] = out.cpu().numpy()
    from util import flat_grad_dict
    for k, v in flat_grad_dict(model).items():
        t["g/" + k] = v
    res.append(t)
for k in res[0]:
    a, b = res[0][k], res[1][k]
    nd = int((a != b).sum())
    print(k, "n", a.size, "differ", nd, "max abs", float(np.abs(a - b).max()), "rel_l2", float(np.sqrt(((a - b) ** 2).sum() / max((b ** 2).sum(), 1e-30))))
