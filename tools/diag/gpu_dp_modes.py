"""Diagnostic (GPU box, one rank): step time of the data-parallel variants of fused_step with a world-1 RCCL group."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import torch.distributed as dist
from argparse import Namespace
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29574")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
from torch_vae_amd.models import VanillaVAE
from torch_vae_amd.train import SyntheticPianorollLoader, build_optimizer, fused_step, enable_library_allreduce, _allreduce_range
H, L, B = 128, 16, 256
cfg = Namespace(batch_size_per_gpu=B, world_size=1, lr_relative=0.01, weight_decay=0.0, optimizer="AdamW", scheduler="OneCycle", epochs=1, freeze_encoder=False)
x = SyntheticPianorollLoader(B, H, 1, device="cuda").batch(0)[0]
scheds = []
def build(lib):
    model = VanillaVAE(1, L, H, generalised=True, compute_dtype="bf16", max_batch=B).cuda()
    opt, sched = build_optimizer(cfg, model, steps_per_epoch=100000)
    scheds.append(sched)
    if lib:
        assert enable_library_allreduce(model)
    return model, opt
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
m, o = build(False)
def plain():
    m.fused_forward_backward(x); o.step()
run("no all-reduce", plain)
def split_only():
    m.fused_forward_backward(x, on_decoder_grads=lambda: None); o.step()
run("no all-reduce, backward split in two calls", split_only)
def torch_inline():
    m.fused_forward_backward(x); _allreduce_range(m, "decoder"); _allreduce_range(m, "encoder"); o.step()
run("torch.distributed, two in-line ops after the backward", torch_inline)
m2, o2 = build(True)
run("library comm, overlap (decoder bucket on the comm stream)", lambda: fused_step(m2, o2, x, overlap=True))
run("library comm, one group after the backward", lambda: fused_step(m2, o2, x, overlap=False))
run("library comm, overlap, + scheduler step", lambda: (fused_step(m2, o2, x, overlap=True), scheds[1].step()))
pool = SyntheticPianorollLoader(B, H, n_batches=4, seed=0, device="cuda", pool=4)
bs = [pool.batch(i)[0] for i in range(4)]
cnt = [0]
def rot():
    cnt[0] += 1
    fused_step(m2, o2, bs[cnt[0] % 4], overlap=True); scheds[1].step()
run("library comm, overlap, + scheduler step, rotating batches", rot)
m2.eps_seed = 7919
run("... + eps_seed set", rot)
run("library comm, overlap (again)", lambda: fused_step(m2, o2, x, overlap=True))
print("library comm world:", m2.library_comm_world())
dist.destroy_process_group()
