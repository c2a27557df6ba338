"""Diagnostic (GPU box): host enqueue time vs GPU time per step, with and without a (single-rank) process group."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist
from argparse import Namespace
from torch_vae_amd.models import VanillaVAE
from torch_vae_amd.train import SyntheticPianorollLoader, build_optimizer, fused_step
H, L, B = 128, 16, 256
model = VanillaVAE(1, L, H, generalised=True, compute_dtype="bf16", max_batch=B).cuda()
cfg = Namespace(batch_size_per_gpu=B, world_size=1, lr_relative=0.01, weight_decay=0.0, optimizer="AdamW", scheduler="OneCycle", epochs=1, freeze_encoder=False)
opt, sched = build_optimizer(cfg, model, steps_per_epoch=100000)
x = SyntheticPianorollLoader(B, H, 1, device="cuda").batch(0)[0]
def run(tag):
    for _ in range(5):
        fused_step(model, opt, x); sched.step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(30):
        fused_step(model, opt, x); sched.step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"{tag}: host enqueue {1e3*(t1-t0)/30:.3f} ms/step, total {1e3*(t2-t0)/30:.3f} ms/step", flush=True)
run("no process group")
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29573")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
run("single-rank RCCL group")
dist.destroy_process_group()
