"""Diagnostic (GPU box): per-wave cycle shares of the producer / consumer weight-gradient kernel (wgrad_split.cuh).
Needs the stamps build; run as:  VAE_STEP_LIB=torch_vae_amd/lib/libvae_step_stamps.so python tools/diag/gpu_stamps_wgrad.py ['{"option": value}']"""
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
Lb = _lib.lib()
opts = json.loads(sys.argv[1]) if len(sys.argv) > 1 else {}
for k, v in opts.items():
    assert Lb.vae_set_option(model._ctx.handle, k.encode(), v) == 0
names = ["prologue", "G part / MFMA work", "barrier wait", "S part (first use of the prefetched loads)", "-", "total"]
for tag in ("encoder.2", "encoder.3", "decoder.0", "decoder.1"):
    buf = torch.zeros(512 * 16 * 8, dtype=torch.int64, device="cuda")
    Lb.vae_debug_stamps(model._ctx.handle, tag.encode(), 32, buf.data_ptr())
    model.fused_forward_backward(x); torch.cuda.synchronize()
    Lb.vae_debug_stamps(model._ctx.handle, b"", 0, None)
    t = buf.view(-1, 16, 8).double()
    t = t[t[:, :, 5].sum(1) > 0]
    for role, sl in (("consumers (waves 0-7)", slice(0, 8)), ("producers (waves 8-15)", slice(8, 16))):
        m = t[:, sl, :].mean((0, 1))
        print(f"{tag} {opts} {role}: workgroups {t.shape[0]} ->", {n: f"{m[k].item():.0f}" for k, n in enumerate(names) if n != "-"}, flush=True)
