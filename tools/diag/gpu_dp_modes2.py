"""Diagnostic: library-comm overlap with ONE model in the process (as bench.py)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import torch.distributed as dist
from argparse import Namespace
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29574")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
from torch_vae_amd.models import VanillaVAE
from torch_vae_amd.train import SyntheticPianorollLoader, build_optimizer, fused_step, enable_library_allreduce
H, L, B = 128, 16, 256
cfg = Namespace(batch_size_per_gpu=B, world_size=1, lr_relative=0.01, weight_decay=0.0, optimizer="AdamW", scheduler="OneCycle", epochs=1, freeze_encoder=False)
x = SyntheticPianorollLoader(B, H, 1, device="cuda").batch(0)[0]
model = VanillaVAE(1, L, H, generalised=True, compute_dtype="bf16", max_batch=B).cuda()
opt, sched = build_optimizer(cfg, model, steps_per_epoch=100000)
assert enable_library_allreduce(model)
def run(tag, step):
    for _ in range(8):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(40):
        step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"{tag}: host enqueue {1e3*(t1-t0)/40:.3f} ms/step, total {1e3*(t2-t0)/40:.3f} ms/step", flush=True)
mode = sys.argv[1] if len(sys.argv) > 1 else "a"
if mode == "a":
    run("overlap first", lambda: fused_step(model, opt, x, overlap=True))
    run("in-line group", lambda: fused_step(model, opt, x, overlap=False))
    run("overlap again", lambda: fused_step(model, opt, x, overlap=True))
else:
    run("in-line group first", lambda: fused_step(model, opt, x, overlap=False))
    run("overlap", lambda: fused_step(model, opt, x, overlap=True))
    run("in-line group", lambda: fused_step(model, opt, x, overlap=False))
dist.destroy_process_group()
