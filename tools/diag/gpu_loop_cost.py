"""Diagnostic (GPU box): where the drop-in loop's time goes.  Per-step time of train_one_epoch fed float32 host batches, uint8 host
batches, bit-plane host batches (train.pack_bits) and device-resident float32 batches, against the bare fused step."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from argparse import Namespace
from torch_vae_amd.models import VanillaVAE
from torch_vae_amd.train import SyntheticPianorollLoader, build_optimizer, fused_step, train_one_epoch, pack_bits
H, L, B, nb = 128, 16, 256, 40
cfg = Namespace(batch_size_per_gpu=B, world_size=1, lr_relative=0.01, weight_decay=0.0, optimizer="AdamW", scheduler="OneCycle", epochs=1,
                log_wandb=False, print_interval=100000, log_interval=100000, freeze_encoder=False, global_rank=0)
model = VanillaVAE(1, L, H, generalised=True, compute_dtype="bf16", max_batch=B).cuda()
opt, sched = build_optimizer(cfg, model, steps_per_epoch=100000)
dev = [SyntheticPianorollLoader(B, H, 1, seed=i, device="cuda").batch(0)[0] for i in range(4)]
lab = torch.zeros(B, dtype=torch.long)
host32 = [(d.cpu().pin_memory(), lab) for d in dev]
host8 = [(d.cpu().to(torch.uint8).pin_memory(), lab) for d in dev]
hostb = [(pack_bits(d.cpu()).pin_memory(), lab) for d in dev]
devl = [(d, lab.cuda()) for d in dev]
def run(tag, loader):
    import io, contextlib
    with contextlib.redirect_stdout(io.StringIO()):
        train_one_epoch(cfg, model, opt, sched, model.loss, [loader[i % 4] for i in range(8)], device="cuda", epoch=2)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        train_one_epoch(cfg, model, opt, sched, model.loss, [loader[i % 4] for i in range(nb)], device="cuda", epoch=2)
        torch.cuda.synchronize()
    print(f"{tag}: {1e3 * (time.perf_counter() - t0) / nb:.3f} ms/step", flush=True)
for _ in range(8): fused_step(model, opt, dev[0])
torch.cuda.synchronize(); t0 = time.perf_counter()
for i in range(nb): fused_step(model, opt, dev[i % 4]); sched.step()
torch.cuda.synchronize(); print(f"bare fused_step + scheduler: {1e3 * (time.perf_counter() - t0) / nb:.3f} ms/step", flush=True)
for rep in range(2):
    run("train_one_epoch, device-resident float32 batches", devl)
    run("train_one_epoch, bit-plane host batches", hostb)
    run("train_one_epoch, uint8 host batches", host8)
    run("train_one_epoch, float32 host batches (blocking copy)", host32)
