"""Diagnostic (GPU box, one rank): step time of the one-call training step with no gradient exchange, with the in-line RCCL group
(exchange 1) and with the bucketed exchange on the communication stream (exchange 2), on a world-1 RCCL communicator.
Run once per hardware-queue setting:  GPU_MAX_HW_QUEUES=8 python tools/diag/gpu_dp_exchange.py"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import torch.distributed as dist
from argparse import Namespace
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29575")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
from torch_vae_amd.models import VanillaVAE
from torch_vae_amd.train import SyntheticPianorollLoader, build_optimizer, enable_library_allreduce
H, L, B = 128, 16, 256
cfg = Namespace(batch_size_per_gpu=B, world_size=1, lr_relative=0.01, weight_decay=0.0, optimizer="AdamW", scheduler="OneCycle", epochs=1, freeze_encoder=False)
x = SyntheticPianorollLoader(B, H, 1, device="cuda").batch(0)[0]
model = VanillaVAE(1, L, H, generalised=True, compute_dtype="bf16", max_batch=B).cuda()
opt, sched = build_optimizer(cfg, model, steps_per_epoch=100000)
assert enable_library_allreduce(model)
model._context(B)
print("GPU_MAX_HW_QUEUES", os.environ.get("GPU_MAX_HW_QUEUES"), "library comm world", model.library_comm_world(), flush=True)
def run(tag, ex):
    for _ in range(8):
        model.fused_train_step(opt, x, exchange=ex)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(40):
        model.fused_train_step(opt, x, exchange=ex)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"exchange {ex} ({tag}): host enqueue {1e3*(t1-t0)/40:.3f} ms/step, total {1e3*(t2-t0)/40:.3f} ms/step", flush=True)
for rep in range(2):
    run("none", 0); run("one in-line group", 1); run("bucketed on the communication stream", 2)
dist.destroy_process_group()
