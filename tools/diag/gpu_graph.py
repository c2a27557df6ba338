"""Diagnostic (GPU box): capture forward+ELBO+backward into a HIP graph (torch.cuda.CUDAGraph) and compare replay with eager."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from argparse import Namespace
from torch_vae_amd.models import VanillaVAE
from torch_vae_amd.train import SyntheticPianorollLoader, build_optimizer
H, L, B = (int(sys.argv[1]), 16, int(sys.argv[2])) if len(sys.argv) > 2 else (128, 16, 256)
model = VanillaVAE(1, L, H, generalised=(H != 32), compute_dtype="bf16", max_batch=B).cuda()
cfg = Namespace(batch_size_per_gpu=B, world_size=1, lr_relative=0.01, weight_decay=0.0, optimizer="AdamW", scheduler="OneCycle", epochs=1, freeze_encoder=False)
opt, sched = build_optimizer(cfg, model, steps_per_epoch=100000)
x = SyntheticPianorollLoader(B, H, 1, device="cuda").batch(0)[0]
xs = x.clone()
def eager():
    out3, _ = model.fused_forward_backward(xs); opt.step(); sched.step(); return out3
for _ in range(5): eager()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(50): eager()
torch.cuda.synchronize(); print(f"eager : {(time.perf_counter()-t0)/50*1e3:.3f} ms/step", flush=True)
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3): model.fused_forward_backward(xs)
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
with torch.cuda.graph(g):
    out3_static, _ = model.fused_forward_backward(xs)
torch.cuda.synchronize()
def graphed():
    g.replay(); opt.step(); sched.step(); return out3_static
ref = eager().clone()
a = graphed().clone()
print("loss eager/graph:", ref.tolist(), a.tolist(), flush=True)
for _ in range(5): graphed()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(50): graphed()
torch.cuda.synchronize(); print(f"graph : {(time.perf_counter()-t0)/50*1e3:.3f} ms/step", flush=True)
