"""Diagnostic (GPU box): per-tensor gradient gaps of a 16-bit mode against the exact and the storage-emulating oracle.
python tools/diag/gpu_emu_gaps.py H L B dtype [gen]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from oracle import vae_oracle as vo
from tests.util import flat_grad_dict, make_model, perturbed_params, rel_l2, PRE_BN_BIAS
H, L, B, dtype = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
gen = (sys.argv[5] == "1") if len(sys.argv) > 5 else H != 32
p = perturbed_params(L, H, 17, gen)
x = vo.synth_pianoroll(B, H, 3); eps = vo.counter_normal(B * L, 3, 5).reshape(B, L)
m = make_model(H, L, gen, dtype, p)
m.fused_forward_backward(torch.from_numpy(x).cuda(), eps=torch.from_numpy(eps).float().cuda())
got = flat_grad_dict(m)
c = vo.forward(p, x.astype(np.float64), eps, None, train=True); g = vo.backward(p, c)
vo.F16_FLUSH_SUBNORMALS = os.environ.get("EMU_F16_FLUSH") == "1"
ce = vo.forward(p, x.astype(np.float64), eps, None, train=True, storage=dtype); ge = vo.backward(p, ce)
for n in got:
    if n in PRE_BN_BIAS: continue
    print(f"{n:28s} vs exact {rel_l2(got[n], g[n].reshape(-1)):.3f}   vs emulated {rel_l2(got[n], ge[n].reshape(-1)):.3f}   emulated vs exact {rel_l2(ge[n].reshape(-1), g[n].reshape(-1)):.3f}   |g| {np.linalg.norm(g[n]):.2e}")
