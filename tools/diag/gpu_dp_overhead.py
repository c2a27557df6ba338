"""Diagnostic: process group created BEFORE the model/context (as bench.py does)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import torch.distributed as dist
from argparse import Namespace
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29574")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
from torch_vae_amd.models import VanillaVAE
from torch_vae_amd.train import SyntheticPianorollLoader, build_optimizer, fused_step
H, L, B = 128, 16, 256
model = VanillaVAE(1, L, H, generalised=True, compute_dtype="bf16", max_batch=B).cuda()
cfg = Namespace(batch_size_per_gpu=B, world_size=1, lr_relative=0.01, weight_decay=0.0, optimizer="AdamW", scheduler="OneCycle", epochs=1, freeze_encoder=False)
opt, sched = build_optimizer(cfg, model, steps_per_epoch=100000)
x = SyntheticPianorollLoader(B, H, 1, device="cuda").batch(0)[0]
def run(tag, step):
    for _ in range(5):
        step(); sched.step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(30):
        step(); sched.step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"{tag}: host enqueue {1e3*(t1-t0)/30:.3f} ms/step, total {1e3*(t2-t0)/30:.3f} ms/step", flush=True)
run("group first, fused_step (decoder bucket on the context comm stream, encoder in line)", lambda: fused_step(model, opt, x))
def plain():
    model.fused_forward_backward(x); opt.step()
run("group first, no all-reduce at all", plain)
def one_ar():
    out = model.fused_forward_backward(x)
    off, n = model.group_range("encoder")
    dist.all_reduce(model.flat_grads()[off:off + n], async_op=True).wait()
    opt.step()
run("group first, one all-reduce at the end", one_ar)
def one_ar_sync():
    out = model.fused_forward_backward(x)
    off, n = model.group_range("encoder")
    dist.all_reduce(model.flat_grads()[off:off + n], async_op=False)
    opt.step()
run("group first, one SYNC all-reduce at the end (current stream)", one_ar_sync)
side = torch.cuda.Stream()
def two_ar_sync():
    def hook():
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            off, n = model.group_range("decoder")
            dist.all_reduce(model.flat_grads()[off:off + n], async_op=False)
    model.fused_forward_backward(x, on_decoder_grads=hook)
    off, n = model.group_range("encoder")
    dist.all_reduce(model.flat_grads()[off:off + n], async_op=False)
    torch.cuda.current_stream().wait_stream(side)
    opt.step()
run("group first, decoder bucket on a side stream + encoder bucket in line (sync ops)", two_ar_sync)
dist.destroy_process_group()
