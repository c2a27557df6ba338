"""Diagnostic (GPU box): start/end of every profiled launch of one training step (both streams)."""
import sys, os, json, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from argparse import Namespace
from torch_vae_amd import _lib
from torch_vae_amd.models import VanillaVAE
from torch_vae_amd.train import SyntheticPianorollLoader, build_optimizer
H = int(os.environ.get("VAE_TL_SIZE", "128")); L, B = 16, 256
model = VanillaVAE(1, L, H, generalised=(H != 32), compute_dtype="bf16", max_batch=B).cuda()
cfg = Namespace(batch_size_per_gpu=B, world_size=1, lr_relative=0.01, weight_decay=0.0, optimizer="AdamW", scheduler="OneCycle", epochs=1, freeze_encoder=False)
opt, sched = build_optimizer(cfg, model, steps_per_epoch=100000)
x = SyntheticPianorollLoader(B, H, 1, device="cuda").batch(0)[0]
Lb = _lib.lib()
for k, v in (json.loads(sys.argv[1]) if len(sys.argv) > 1 else {}).items():
    assert Lb.vae_set_option(model._context(B).handle, k.encode(), v) == 0
for _ in range(3):
    model.fused_forward_backward(x); opt.step()
torch.cuda.synchronize()
Lb.vae_profile(model._ctx.handle, 1)
model.fused_forward_backward(x); opt.step()
buf = ctypes.create_string_buffer(1 << 18)
assert Lb.vae_profile_timeline(model._ctx.handle, buf, len(buf)) == 0
Lb.vae_profile(model._ctx.handle, 0)
for name, t0, t1, _bytes in sorted(json.loads(buf.value.decode()), key=lambda r: r[1]):
    print(f"{1e3*t0:8.1f} {1e3*t1:8.1f} {1e3*(t1-t0):7.1f} us  {name}")
