"""Ablation / knob timing helper for the GPU box (not a pytest file): per-kernel HIP-event times under option settings."""
import sys, os, time, json, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from argparse import Namespace
from torch_vae_amd import _lib
from torch_vae_amd.models import VanillaVAE
from torch_vae_amd.train import SyntheticPianorollLoader, build_optimizer
H, L, B = 128, 16, 256
model = VanillaVAE(1, L, H, generalised=True, compute_dtype="bf16", max_batch=B).cuda()
cfg = Namespace(batch_size_per_gpu=B, world_size=1, lr_relative=0.01, weight_decay=0.0, optimizer="AdamW", scheduler="OneCycle", epochs=1, freeze_encoder=False)
opt, sched = build_optimizer(cfg, model, steps_per_epoch=100000)
x = SyntheticPianorollLoader(B, H, 1, device="cuda").batch(0)[0]
model.fused_forward_backward(x); opt.step()
L_ = _lib.lib()
for knob, val in json.loads(sys.argv[1]):
    assert L_.vae_set_option(model._ctx.handle, knob.encode(), val) == 0
    for _ in range(3): model.fused_forward_backward(x)
    L_.vae_profile(model._ctx.handle, 1)
    for _ in range(5): model.fused_forward_backward(x)
    buf = ctypes.create_string_buffer(1 << 16)
    L_.vae_profile_report(model._ctx.handle, buf, len(buf)); L_.vae_profile(model._ctx.handle, 0)
    ks = json.loads(buf.value.decode())
    print(knob, val, "total ms/step", round(sum(k["ms"] for k in ks) / 5, 3))
    for k in sorted(ks, key=lambda k: -k["ms"]):
        if any(s in k["name"] for s in ("down_", "up_")): print(f"   {k['name']:44s} {k['ms']/k['calls']*1e3:7.1f} us")
