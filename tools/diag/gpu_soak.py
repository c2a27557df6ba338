"""Diagnostic (GPU box): a few hundred bf16 training steps at the benchmark size - the ELBO must fall, stay finite and two
identical runs must agree bit for bit (races between the caller's stream and the side streams would break that)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from argparse import Namespace
from torch_vae_amd.models import VanillaVAE
from torch_vae_amd.train import SyntheticPianorollLoader, build_optimizer, fused_step
H, L, B, STEPS = 128, 16, 256, int(sys.argv[1]) if len(sys.argv) > 1 else 300
def run():
    torch.manual_seed(0)
    model = VanillaVAE(1, L, H, generalised=True, compute_dtype="bf16", max_batch=B).cuda()
    cfg = Namespace(batch_size_per_gpu=B, world_size=1, lr_relative=0.01, weight_decay=0.0, optimizer="AdamW", scheduler="OneCycle", epochs=1, freeze_encoder=False)
    opt, sched = build_optimizer(cfg, model, steps_per_epoch=STEPS)
    pool = SyntheticPianorollLoader(B, H, n_batches=8, seed=5, device="cuda", pool=8)
    xs = [pool.batch(i)[0] for i in range(8)]
    model.eps_seed = 1234
    hist = []
    for s in range(STEPS):
        out3, _ = fused_step(model, opt, xs[s % 8]); sched.step()
        if s % 25 == 0 or s == STEPS - 1:
            hist.append(out3.tolist())
    torch.cuda.synchronize()
    return hist, model.flat_parameters().detach().clone()
t0 = time.time(); h1, p1 = run(); t1 = time.time(); h2, p2 = run()
for i, v in enumerate(h1): print(i * 25, [round(x, 5) for x in v])
print("finite:", bool(torch.isfinite(p1).all()), " loss fell:", h1[-1][0] < h1[0][0], " identical runs:", bool(torch.equal(p1, p2)) and h1 == h2, f" ({t1 - t0:.1f} s per run)")
