"""Diagnostic (GPU box): where the HOST time of one training step goes (cProfile over 300 steps)."""
import sys, os, time, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from argparse import Namespace
from torch_vae_amd.models import VanillaVAE
from torch_vae_amd.train import SyntheticPianorollLoader, build_optimizer, fused_step
H = int(sys.argv[1]) if len(sys.argv) > 1 else 32
L, B = 16, 256
model = VanillaVAE(1, L, H, generalised=(H != 32), compute_dtype="bf16", max_batch=B).cuda()
cfg = Namespace(batch_size_per_gpu=B, world_size=1, lr_relative=0.01, weight_decay=0.0, optimizer="AdamW", scheduler="OneCycle", epochs=1, freeze_encoder=False)
opt, sched = build_optimizer(cfg, model, steps_per_epoch=100000)
x = SyntheticPianorollLoader(B, H, 1, device="cuda").batch(0)[0]
def step():
    fused_step(model, opt, x); sched.step()
for _ in range(20): step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(300): step()
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print(f"host {1e3*(t1-t0)/300:.3f} ms/step, total {1e3*(t2-t0)/300:.3f} ms/step")
pr = cProfile.Profile(); pr.enable()
for _ in range(300): step()
pr.disable(); torch.cuda.synchronize()
st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(22)
