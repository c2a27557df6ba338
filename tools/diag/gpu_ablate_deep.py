"""Diagnostic (GPU box, stamps build): isolated time of the deep-layer kernels under the ablation bits of DeepConvArgs::ablate.
VAE_STEP_LIB=torch_vae_amd/lib/libvae_step_stamps.so python tools/diag/gpu_ablate_deep.py"""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from torch_vae_amd import _lib
from torch_vae_amd.models import VanillaVAE
from torch_vae_amd.train import SyntheticPianorollLoader
H, L, B = 128, 16, 256
model = VanillaVAE(1, L, H, generalised=True, compute_dtype="bf16", max_batch=B).cuda()
x = SyntheticPianorollLoader(B, H, 1, device="cuda").batch(0)[0]
model.fused_forward_backward(x)
Lb = _lib.lib(); h = model._ctx.handle
assert Lb.vae_set_option(h, b"use_side_stream", 0) == 0
for ab in (0, 16, 1, 4, 20):
    assert Lb.vae_set_option(h, b"knob_ablate_f", ab) == 0
    for _ in range(3): model.fused_forward_backward(x)
    Lb.vae_profile(h, 1)
    for _ in range(10): model.fused_forward_backward(x)
    buf = (b" " * (1 << 20)); import ctypes
    cb = ctypes.create_string_buffer(1 << 20)
    Lb.vae_profile_report(h, cb, 1 << 20); Lb.vae_profile(h, 0)
    rep = json.loads(cb.value.decode())
    sel = {r["name"]: round(1e3 * r["ms"] / r["calls"], 1) for r in rep if r["name"].startswith(("down_", "up_")) and any(t in r["name"] for t in ("encoder.2", "encoder.3", "decoder.0", "decoder.1"))}
    print("ablate", ab, sel)
