"""Diagnostic (GPU box): per-phase cycle shares of the pipelined down kernel for chosen layers."""
import sys, os, json, ctypes
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
for k, v in (json.loads(sys.argv[1]) if len(sys.argv) > 1 else {}).items():
    assert Lb.vae_set_option(model._ctx.handle, k.encode(), v) == 0
names = ["barrierA", "write_patch", "barrierB", "issue", "mfma", "epilogue", "prologue", "tail"]
for tag, epi in [("encoder.3", 0), ("encoder.2", 0), ("encoder.1", 0), ("decoder.0", 2), ("decoder.1", 1),
                 ("final_layer.0", 16), ("decoder.2", 16), ("decoder.0", 16), ("encoder.1", 17), ("encoder.3", 17)]:   # +16: up kernels
    buf = torch.zeros(1024 * 8 * 24, dtype=torch.int64, device="cuda")
    Lb.vae_debug_stamps(model._ctx.handle, tag.encode(), epi, buf.data_ptr())
    model.fused_forward_backward(x); torch.cuda.synchronize()
    Lb.vae_debug_stamps(model._ctx.handle, b"", 0, None)
    t = buf.view(-1, 6 if epi & 16 else 8).double(); t = t[t.sum(1) > 0]   # down kernels also stamp prologue / tail
    tot = t.sum(1).mean().item()
    print(f"{tag} epi={epi}: waves {t.shape[0]}, cycles/wave {tot:.0f} ->", {n: f"{t[:,k].mean().item():.0f}" for k, n in enumerate(names[:t.shape[1]])})
