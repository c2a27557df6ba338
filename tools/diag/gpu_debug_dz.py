import sys, os, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import vae_oracle as vo
from tests.util import perturbed_params, make_model
from tests.gpu_debug_layers import dbg
H, L, B, gen = 32, 16, 33, False
p = perturbed_params(L, H, 3, gen)
p = vo.init_params(L, H, 3, gen)
rng = np.random.default_rng(3)
for k in p:
    if k.endswith(".1.weight"): p[k] = 1 + 0.2 * rng.standard_normal(p[k].shape)
    if k.endswith(".1.bias") or k.endswith(".0.bias") or k.endswith("3.bias"): p[k] = 0.1 * rng.standard_normal(p[k].shape)
model = make_model(H, L, gen, "f32", p)
x = vo.synth_pianoroll(B, H, 3); eps = vo.counter_normal(B * L, 3, 5).reshape(B, L)
model.fused_forward_backward(torch.from_numpy(x).cuda(), eps=torch.from_numpy(eps).float().cuda())
c = vo.forward(p, x.astype(np.float64), eps, None, train=True); g = vo.backward(p, c)
for i, n in [(7, "final_layer"), (6, "decoder.2")]:
    got = dbg(model, 8 + i, g[n + ".dz"].shape); want = g[n + ".dz"]
    d = np.abs(got - want); idx = np.argsort(d.reshape(-1))[::-1][:6]
    print(n, "rms", np.sqrt((want**2).mean()), "nbad(>1e-3*rms)", int((d > 1e-3*np.sqrt((want**2).mean())).sum()))
    for j in idx:
        u = np.unravel_index(j, want.shape)
        print("  ", u, "got", got[u], "want", want[u], "z", c[n + ".z"][u])
