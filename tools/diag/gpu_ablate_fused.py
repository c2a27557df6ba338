"""Diagnostic (GPU box): isolated time of the encoder.1 fused backward kernel with phases ablated (results are wrong, timing only)."""
import sys, os, json, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from torch_vae_amd import _lib
from torch_vae_amd.models import VanillaVAE
from torch_vae_amd.train import SyntheticPianorollLoader
H, L, B = 128, 16, 256
model = VanillaVAE(1, L, H, generalised=True, compute_dtype="bf16", max_batch=B).cuda()
x = SyntheticPianorollLoader(B, H, 1, device="cuda").batch(0)[0]
Lb = _lib.lib(); h = model._context(B).handle
Lb.vae_set_option(h, b"use_side_stream", 0)
for abl in [0, 1, 2, 4, 8, 3, 7, 12, 15] + [int(a) for a in sys.argv[1:]]:
    Lb.vae_set_option(h, b"knob_ablate_f", abl)
    for _ in range(2):
        model.fused_forward_backward(x)
    torch.cuda.synchronize()
    Lb.vae_profile(h, 1)
    for _ in range(3):
        model.fused_forward_backward(x)
    buf = ctypes.create_string_buffer(1 << 16)
    Lb.vae_profile_report(h, buf, len(buf)); Lb.vae_profile(h, 0)
    ks = {k["name"]: k for k in json.loads(buf.value.decode())}
    k = ks["conv_bwd_fused(dgrad+wgrad) @encoder.1"]
    print(f"ablate {abl:2d}: {1e3 * k['ms'] / k['calls']:7.1f} us")
