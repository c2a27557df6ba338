"""Worker of test_parity_gpu.py::test_data_parallel_product_path_two_ranks (not collected by pytest).

Two ranks share this box's GPU, so the exchange goes through gloo; everything else is the product path:
build_optimizer (initial broadcast, lr scaling by world size), train.fused_step (forward, ELBO, backward, gradient
all-reduce, fused AdamW), OneCycle.  Checked against the oracle's simulation of R replicas: per-replica BatchNorm,
averaged gradients, one AdamW update - the check tests/test_host_cpu.py does for the oracle alone.
"""
import os
import sys
from argparse import Namespace

import numpy as np
import torch
import torch.distributed as dist

ROOT = sys.argv[1]
sys.path.insert(0, ROOT)
from oracle import vae_oracle as vo  # noqa: E402
from tests.util import PRE_BN_BIAS, load_params, rel_l2  # noqa: E402
from torch_vae_amd import _lib  # noqa: E402
from torch_vae_amd.models import VanillaVAE  # noqa: E402
from torch_vae_amd.train import build_optimizer, fused_step  # noqa: E402

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
torch.cuda.set_device(0)
H, L, B, STEPS, TOTAL, SEED = 32, 16, 4, 3, 10, 21

# deliberately different replicas: rank 0 holds the oracle's weights, the others torch's own random init and perturbed
# BatchNorm buffers; build_optimizer must make them identical
torch.manual_seed(1234 + rank)
model = VanillaVAE(1, L, H, compute_dtype="f32").to("cuda")
if rank == 0:
    load_params(model, vo.init_params(L, H, SEED))
else:
    model._bnflat.add_(0.5)
cfg = Namespace(batch_size_per_gpu=B, world_size=world, lr_relative=0.01, weight_decay=0.0, optimizer="AdamW", scheduler="OneCycle",
                epochs=1, freeze_encoder=False)
opt, sched = build_optimizer(cfg, model, steps_per_epoch=TOTAL)
assert abs(cfg.lr - 0.01 * B * world / 128) < 1e-15                      # train.py:165-166, 201
flat0 = model.flat_parameters().detach().clone()
ref = flat0.clone(); dist.broadcast(ref, 0)
assert torch.equal(flat0, ref), "initial broadcast did not make the replicas identical"
assert float(model._bnflat.abs().max()) == 1.0                           # rank 0's fresh running statistics everywhere

# device-side reparameterisation noise must differ between ranks (the rank is mixed into the seed)
model._run_forward(torch.from_numpy(vo.synth_pianoroll(B, H, 1)).cuda(), None, train=False)
e = torch.empty(B, L, device="cuda")
_lib.check(_lib.lib().vae_last_eps(model._ctx.handle, e.data_ptr(), torch.cuda.current_stream().cuda_stream), "eps")
e = e.cpu()
both = [torch.empty_like(e) for _ in range(world)]
dist.all_gather(both, e)
assert not torch.equal(both[0], both[1]), "ranks drew identical reparameterisation noise"

inputs = lambda r, s: (vo.synth_pianoroll(B, H, 500 + r + 10 * s), vo.counter_normal(B * L, 500 + r + 10 * s, 5).reshape(B, L))  # noqa: E731
losses = []
for s in range(STEPS):
    x, eps = inputs(rank, s)
    out3, _ = fused_step(model, opt, torch.from_numpy(x).cuda(), eps=torch.from_numpy(eps).float().cuda())
    sched.step()
    losses.append(out3.tolist())
    flat = model.flat_parameters().detach().clone()
    ref = flat.clone(); dist.broadcast(ref, 0)
    assert torch.equal(flat, ref), f"replicas diverged at step {s}"
    # .grad holds the MEAN over replicas (what a caller of the reference's single-process loop would see at the global batch)
    for prefix in ("encoder", "decoder"):
        off, n = model.group_range(prefix)
        g = model.flat_grads()[off:off + n].detach().clone(); gr = g.clone(); dist.broadcast(gr, 0)
        assert torch.equal(g, gr), f"{prefix} gradients differ between ranks after the exchange"

# oracle: R replicas simulated in one process (f64)
tr = vo.make_trainer(L, H, B, TOTAL, seed=SEED, world_size=world)
want_losses = []
for s in range(STEPS):
    sims, lo_r = [], []
    for r in range(world):
        x, eps = inputs(r, s)
        c = vo.forward(tr.p, x.astype(np.float64), eps, None, train=True)
        lo = vo.loss(c)
        lo_r.append([float(lo["loss"]), float(lo["reconstruction_loss"]), float(lo["kld_loss"])])
        sims.append(vo.backward(tr.p, c))
    mean = {k: sum(g_[k] for g_ in sims) / world for k in sims[0] if not k.startswith("__") and not k.endswith(".dz")}
    x, eps = inputs(rank, s)
    tr.step(x.astype(np.float64), eps, grads_override=lambda _g: mean)
    want_losses.append(lo_r[rank])
np.testing.assert_allclose(np.array(losses), np.array(want_losses), rtol=3e-4)
sd = model.state_dict()
worst = 0.0
for k, v in tr.p.items():
    if k in PRE_BN_BIAS:
        continue
    worst = max(worst, rel_l2(sd[k].cpu().numpy(), v))
assert worst < 2e-4, worst
print(f"DP_PRODUCT_OK rank {rank} worst param rel-L2 {worst:.2e}")
dist.destroy_process_group()
