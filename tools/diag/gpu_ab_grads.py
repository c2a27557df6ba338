"""Diagnostic (GPU box): per-tensor gradient differences between two option sets on one case.
python tools/diag/gpu_ab_grads.py H L B dtype 'optA=0,optB=1' 'optA=1'"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from oracle import vae_oracle as vo
from tests.util import flat_grad_dict, make_model, perturbed_params, rel_l2, PRE_BN_BIAS
from torch_vae_amd import _lib
H, L, B, dtype = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
gen = H != 32
p = perturbed_params(L, H, 17, gen)
x = vo.synth_pianoroll(B, H, 3); eps = vo.counter_normal(B * L, 3, 5).reshape(B, L)
c = vo.forward(p, x.astype(np.float64), eps, None, train=True); g = vo.backward(p, c)
res = []
for opts in sys.argv[5:7]:
    m = make_model(H, L, gen, dtype, p); m._context(B)
    for kv in filter(None, opts.split(",")):
        k, v = kv.split("="); assert _lib.lib().vae_set_option(m._ctx.handle, k.encode(), int(v)) == 0
    m.fused_forward_backward(torch.from_numpy(x).cuda(), eps=torch.from_numpy(eps).float().cuda())
    res.append(flat_grad_dict(m))
for n in res[0]:
    if n in PRE_BN_BIAS: continue
    print(f"{n:28s} A-vs-oracle {rel_l2(res[0][n], g[n].reshape(-1)):.2e}  B-vs-oracle {rel_l2(res[1][n], g[n].reshape(-1)):.2e}  A-vs-B {rel_l2(res[0][n], res[1][n]):.2e}")
