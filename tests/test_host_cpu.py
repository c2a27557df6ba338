"""CPU-only tests: the C-ABI library loads and exports every symbol include/vae_step.h declares,
host-side layout logic, optimiser/scheduler construction, and the data-parallel reduction path
(world_size 2 over gloo).  No kernel is launched here."""
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from torch_vae_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "vae_step.h")).read()
    declared = set(re.findall(r"\b(vae_[a-z0-9_]+)\s*\(", hdr)) - {"vae_ctx"}
    assert declared, "no declarations parsed"
    L = _lib.lib()
    missing = [n for n in sorted(declared) if not hasattr(L, n)]
    assert not missing, missing
    assert set(_lib.EXPORTS) == declared
    assert L.vae_abi_version() == 1


def test_param_layout_matches_reference_census():
    from torch_vae_amd import _lib
    from oracle import vae_oracle as vo
    offs, sizes, total = _lib.param_layout(32, 16, False)
    assert sum(sizes) == 836353                      # SURVEY.md 8a parameter census
    shapes = vo.param_shapes(16, 32)
    assert list(shapes) == list(_lib.PARAM_NAMES)
    assert [int(np.prod(s)) for s in shapes.values()] == sizes
    assert all(o % 64 == 0 for o in offs) and total >= offs[-1] + sizes[-1]
    _, sizes_g, _ = _lib.param_layout(128, 16, True)
    assert sum(sizes_g) == 1588993                   # SURVEY.md 8d, G(128,16)
    with pytest.raises(ValueError):
        _lib.param_layout(64, 16, False)             # the reference-exact model only exists at 32x32
    with pytest.raises(ValueError):
        _lib.param_layout(48, 16, True)


def test_model_mirror_surface_and_state_dict_keys():
    from torch_vae_amd.models import VanillaVAE
    hd = [32, 64, 128, 256]
    m = VanillaVAE(1, 16, 32, hidden_dims=hd, kld_weight=2.0)
    assert hd == [256, 128, 64, 32]                  # models.py:60 mutates the caller's list
    assert m.name == "VanillaVAE" and m.latent_dim == 16 and m.kld_weight == 2.0
    sd = m.state_dict()
    from oracle import vae_oracle as vo
    want = set(vo.param_shapes(16, 32)) | set(vo.init_bn_state())
    assert set(sd) == want
    assert sum(p.numel() for p in m.encoder.parameters()) == 388800
    assert sum(p.numel() for p in m.decoder.parameters()) == 387744
    # reference init rules (models.py:227-236): xavier conv weights + zero bias in encoder/final conv
    assert float(m.encoder[0][0].bias.abs().max()) == 0.0
    assert float(m.final_layer[3].bias.abs().max()) == 0.0
    assert float(m.decoder[0][0].bias.abs().max()) > 0.0   # ConvTranspose2d keeps torch's default init


def test_optimizer_groups_and_onecycle_like_reference():
    from argparse import Namespace
    from torch_vae_amd.models import VanillaVAE
    from torch_vae_amd.train import build_optimizer
    from oracle import vae_oracle as vo
    m = VanillaVAE(1, 16, 32)
    cfg = Namespace(batch_size_per_gpu=32, world_size=2, lr_relative=0.01, weight_decay=0.0, optimizer="AdamW",
                    scheduler="OneCycle", epochs=2, freeze_encoder=False)
    opt, sched = build_optimizer(cfg, m, steps_per_epoch=50)
    assert cfg.batch_size == 64 and abs(cfg.lr - 0.01 * 64 / 128) < 1e-12       # train.py:165-166,201
    assert [g["name"] for g in opt.param_groups] == ["encoder", "decoder"]      # train.py:210-225
    n_opt = sum(p.numel() for g in opt.param_groups for p in g["params"])
    assert n_opt == 776544
    oc = vo.OneCycle(cfg.lr, 100)
    lr0, b0 = oc.value(0)
    assert abs(opt.param_groups[0]["lr"] - lr0) < 1e-12 and abs(opt.param_groups[0]["betas"][0] - b0) < 1e-12


def test_cpu_forward_fails_loudly():
    from torch_vae_amd.models import VanillaVAE
    m = VanillaVAE(1, 16, 32)
    with pytest.raises(RuntimeError, match="no CPU path"):
        m(torch.zeros(2, 1, 32, 32))


def test_roofline_accounting_matches_survey():
    from torch_vae_amd.models import algorithmic_bytes_per_step, count_flops_per_sample
    assert abs(count_flops_per_sample(32, 16, False) - 59.1e6) < 0.1e6
    assert abs(count_flops_per_sample(128, 16, True) - 946.1e6) < 0.2e6
    assert abs(algorithmic_bytes_per_step(32, 16, 256, 2, False) - 196e6) < 1e6
    assert abs(algorithmic_bytes_per_step(128, 16, 256, 2, True) - 2.72e9) < 0.01e9


_DP_WORKER = r"""
import os, sys
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
from oracle import vae_oracle as vo
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
H, L, B = 32, 16, 4
# R replicas: per-replica BatchNorm, averaged gradients, identical AdamW update (SURVEY.md 8e)
p = vo.init_params(L, H, seed=21, dtype=np.float64)
tr = vo.make_trainer(L, H, B, 10, seed=21, world_size=world)
x = vo.synth_pianoroll(B, H, 500 + rank).astype(np.float64)
eps = vo.counter_normal(B * L, 500 + rank, 5).reshape(B, L)

def allreduce_mean(g):
    out = {}
    for k in sorted(g):
        if k.startswith("__") or k.endswith(".dz"):
            continue
        t = torch.from_numpy(np.ascontiguousarray(g[k]))
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        out[k] = (t / world).numpy()
    return out

lo, c, g = tr.step(x, eps, grads_override=allreduce_mean)
# every rank must hold identical parameters after the step
flat = torch.from_numpy(np.concatenate([tr.p[k].reshape(-1) for k in sorted(tr.p)]))
ref = flat.clone(); dist.broadcast(ref, 0)
assert torch.equal(flat, ref), "replicas diverged"
# and they must equal a single-process simulation of both replicas
if rank == 0:
    sims = []
    for r in range(world):
        t2 = vo.make_trainer(L, H, B, 10, seed=21, world_size=world)
        xr = vo.synth_pianoroll(B, H, 500 + r).astype(np.float64); er = vo.counter_normal(B * L, 500 + r, 5).reshape(B, L)
        cc = vo.forward(t2.p, xr, er, t2.bn_state, train=True); sims.append(vo.backward(t2.p, cc))
    mean = {k: sum(s[k] for s in sims) / world for k in sims[0] if not k.startswith("__") and not k.endswith(".dz")}
    t3 = vo.make_trainer(L, H, B, 10, seed=21, world_size=world)
    t3.step(x, eps, grads_override=lambda _g: mean)
    for k in t3.p:
        np.testing.assert_allclose(tr.p[k], t3.p[k], rtol=1e-12, atol=1e-15)
    assert abs(tr.sched_enc.max_lr - 0.01 * B * world / 128) < 1e-15   # train.py:201 lr scaling by world size
    print("DP_OK")
dist.destroy_process_group()
"""


def test_data_parallel_gradient_average_world2_gloo(tmp_path):
    """The N>1 path: sum-all-reduce of gradients / world, per-replica BN, identical update on every
    rank (what torch_vae_amd.train.allreduce_gradients + FusedAdamW.grad_scale do with RCCL)."""
    script = tmp_path / "dp_worker.py"
    script.write_text(_DP_WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29541", OMP_NUM_THREADS="2")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                        "--master-addr", "127.0.0.1", "--master-port", "29541", str(script), ROOT],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    assert "DP_OK" in r.stdout


def test_allreduce_helper_ranges_cover_optimised_groups():
    from torch_vae_amd.models import VanillaVAE
    m = VanillaVAE(1, 16, 32)
    (eo, en), (do, dn) = m.group_range("encoder"), m.group_range("decoder")
    assert eo == 0 and en >= 388800 and dn >= 387744 and do > eo + en
    with pytest.raises(ValueError):
        m.group_range("nonexistent")


_DP_HOST_WORKER = r"""
import os, sys
import torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
from torch_vae_amd import train
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)

class FlatModel:
    # the flat-buffer surface of VanillaVAE that the data-parallel host code touches (no kernels on a CPU box)
    def __init__(self):
        g = torch.Generator().manual_seed(100 + rank)
        self._flat = torch.randn(64, generator=g); self._bnflat = torch.randn(8, generator=g); self._nbt = torch.full((8,), rank, dtype=torch.int64)
        self._g = torch.arange(64, dtype=torch.float32) * (rank + 1)
    def flat_grads(self): return self._g
    def flat_parameters(self): return self._flat
    def group_range(self, prefix): return {"encoder": (0, 16), "decoder": (32, 16)}[prefix]
    def library_comm_world(self): return 0
    def comm_stream(self): raise AssertionError("no stream hand-off on the in-line path")

m = FlatModel()
train.sync_initial_state(m)
ref = [t.clone() for t in (m._flat, m._bnflat, m._nbt)]
for t in ref: dist.broadcast(t, 0)
assert all(torch.equal(a, b) for a, b in zip(ref, (m._flat, m._bnflat, m._nbt))) and int(m._nbt[0]) == 0
train.allreduce_gradients(m)
base = torch.arange(64, dtype=torch.float32)
want = base * (1 + 2) / 2            # mean of rank 0's (x1) and rank 1's (x2) gradients
assert torch.equal(m._g[0:16], want[0:16]) and torch.equal(m._g[32:48], want[32:48])
assert torch.equal(m._g[16:32], base[16:32] * (rank + 1)) and torch.equal(m._g[48:], base[48:] * (rank + 1))   # never-optimised ranges stay local
print("DP_HOST_OK")
dist.destroy_process_group()
"""


def test_data_parallel_host_logic_world2_gloo(tmp_path):
    """torch_vae_amd.train's own data-parallel host code with world_size 2 over gloo on CPU: the initial broadcast makes the
    replicas identical, the exchange leaves the MEAN of the optimised ranges in the gradient buffer and touches nothing else."""
    script = tmp_path / "dp_host_worker.py"
    script.write_text(_DP_HOST_WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29543", OMP_NUM_THREADS="2")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                        "--master-addr", "127.0.0.1", "--master-port", "29543", str(script), ROOT],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    assert r.stdout.count("DP_HOST_OK") == 2


def test_checkpoint_helpers_format_and_atomicity(tmp_path):
    """utils.safe_save_model keeps the reference's FORMAT (utils.py:311-351: one dict of state_dicts + extras + config) and
    never leaves a partial file; should_save reproduces the reference's rank gate (train.py:444) unless told to fix it."""
    from argparse import Namespace
    from torch_vae_amd import utils
    lin = torch.nn.Linear(3, 2)
    cfg = Namespace(checkpoint_path=str(tmp_path / "a" / "b" / "ckpt.pt"), model_output_dir=str(tmp_path), global_rank=0)
    path = utils.safe_save_model({"encoder": lin}, config=cfg, epoch=7, total_step=11)
    assert path == cfg.checkpoint_path and os.listdir(os.path.dirname(path)) == ["ckpt.pt"]      # no temporary left behind
    ck = torch.load(path, weights_only=False)
    assert set(ck) == {"encoder", "epoch", "total_step", "config"} and ck["epoch"] == 7 and torch.equal(ck["encoder"]["weight"], lin.weight)

    class Boom(torch.nn.Module):
        def state_dict(self, *a, **k):
            return {"w": (lambda: 0)}            # unpicklable: torch.save raises mid-write
    with pytest.raises(Exception):
        utils.safe_save_model({"encoder": Boom()}, path)
    assert os.listdir(os.path.dirname(path)) == ["ckpt.pt"] and torch.load(path, weights_only=False)["epoch"] == 7   # old file intact
    with pytest.raises(ValueError):
        utils.safe_save_model({"encoder": lin})
    assert utils.should_save(cfg) is False and utils.should_save(cfg, fix_rank_gate=True) is True      # rank 0 never saves in the reference
    cfg.global_rank = 1
    assert utils.should_save(cfg) is True and utils.should_save(cfg, fix_rank_gate=True) is False
    cfg.model_output_dir = None
    assert utils.should_save(cfg) is False


def test_bit_plane_stimuli_and_overlap_refusal(monkeypatch):
    """Host logic of the training loop that needs no GPU: bit-plane packing / expansion of 0/1 pianorolls (pack_bits and the loop's
    expansion are inverse, in numpy.packbits order), and the loud refusal of the overlapped gradient exchange when eight HIP
    hardware queues are configured (the 2.5x-slower combination)."""
    from torch_vae_amd import train
    g = torch.Generator().manual_seed(0)
    x = (torch.rand(3, 1, 16, 32, generator=g) < 0.3).float()
    packed = train.pack_bits(x)
    assert packed.dtype == torch.uint8 and packed.shape == (3, 1, 16, 4)
    np.testing.assert_array_equal(packed.numpy(), np.packbits(x.numpy().astype(np.uint8), axis=-1))
    assert torch.equal(train._expand_stimuli(packed, 32), x)
    assert torch.equal(train._expand_stimuli(x.to(torch.uint8), 32), x) and torch.equal(train._expand_stimuli(x.bool(), 32), x)
    assert train._expand_stimuli(x, 32) is x
    with pytest.raises(ValueError):
        train.pack_bits(torch.zeros(1, 1, 4, 12))
    monkeypatch.setenv("GPU_MAX_HW_QUEUES", "8")
    with pytest.raises(RuntimeError, match="GPU_MAX_HW_QUEUES=8"):
        train._refuse_overlap_on_eight_queues()
    monkeypatch.setenv("GPU_MAX_HW_QUEUES", "6")
    train._refuse_overlap_on_eight_queues()
    monkeypatch.delenv("GPU_MAX_HW_QUEUES")
    train._refuse_overlap_on_eight_queues()          # HIP's own default is four
