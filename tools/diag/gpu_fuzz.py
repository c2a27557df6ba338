"""Diagnostic (GPU box): a sweep of unusual shapes through the f32 and bf16 paths against the numpy oracle
(small cases) - batch 1, odd batches, tiny / large latent sizes, every supported image size."""
import sys, os, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import vae_oracle as vo
from tests.util import PRE_BN_BIAS, flat_grad_dict, make_model, perturbed_params, rel_l2

cfgs = [(32, 16, 1, False), (32, 4, 3, False), (32, 16, 130, False), (32, 128, 7, True), (64, 8, 1, True), (64, 16, 9, True),
        (64, 128, 5, True), (128, 16, 1, True), (128, 4, 3, True), (256, 16, 1, True), (256, 8, 2, True), (32, 16, 257, False)]
bad = 0
for (H, L, B, gen) in cfgs:
    p = perturbed_params(L, H, 40 + B, gen)
    x = vo.synth_pianoroll(B, H, 60 + B)
    eps = vo.counter_normal(B * L, 60 + B, 5).reshape(B, L)
    t0 = time.time()
    c = vo.forward(p, x.astype(np.float64), eps, None, train=True); lo = vo.loss(c); g = vo.backward(p, c)
    want = np.array([float(lo["loss"]), float(lo["reconstruction_loss"]), float(lo["kld_loss"])])
    for dtype in ("f32", "bf16", "f16"):
        m = make_model(H, L, gen, dtype, p)
        out3, xhat = m.fused_forward_backward(torch.from_numpy(x).cuda(), eps=torch.from_numpy(eps).float().cuda())
        got = np.array(out3.tolist())
        el = np.abs(got - want) / np.maximum(np.abs(want), 1e-12)
        gd = flat_grad_dict(m)
        worst = ("", 0.0)
        for n, v in gd.items():
            if n in PRE_BN_BIAS:
                continue
            ref = g[n].reshape(-1)
            if dtype == "f32":
                e = rel_l2(v, ref)
            else:
                e = 1.0 - float(np.dot(v.astype(np.float64), ref) / (np.linalg.norm(v) * np.linalg.norm(ref) + 1e-30))
            if e > worst[1]:
                worst = (n, e)
        lim_l, lim_g = (1e-4, 5e-3) if dtype == "f32" else (1e-2, 0.03) if dtype == "bf16" else (2e-3, 0.01)
        flag = "" if (el.max() < lim_l and worst[1] < lim_g and np.isfinite(got).all()) else "   <-- CHECK"
        bad += bool(flag)
        print(f"H={H} L={L} B={B} gen={gen} {dtype}: ELBO rel err {el.max():.2e}; worst grad {worst[0]} {worst[1]:.2e}{flag}", flush=True)
    print(f"   (oracle {time.time() - t0:.1f} s)", flush=True)
print("cases to check:", bad)
