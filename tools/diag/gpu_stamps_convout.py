"""Diagnostic (GPU box): per-wave cycle shares of the row-streaming output-conv kernel (convout_stream.cuh).
Needs the stamps build:  make -C torch_vae_amd/csrc STAMPS=1 OUT=../lib/libvae_step_stamps.so OBJD=../lib/obj_stamps
run as:  VAE_STEP_LIB=torch_vae_amd/lib/libvae_step_stamps.so python tools/diag/gpu_stamps_convout.py ['{"option": value}']"""
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
for k, v in (json.loads(sys.argv[1]) if len(sys.argv) > 1 else {}).items():
    assert Lb.vae_set_option(model._ctx.handle, k.encode(), v) == 0
names = ["prologue", "phase1 work", "barrier A", "phase2 work", "copies wait + barrier B", "total"]
buf = torch.zeros(256 * 16 * 8, dtype=torch.int64, device="cuda")
Lb.vae_debug_stamps(model._ctx.handle, b"final_layer.3", 0, buf.data_ptr())
model.fused_forward_backward(x); torch.cuda.synchronize()
Lb.vae_debug_stamps(model._ctx.handle, b"", 0, None)
t = buf.view(-1, 16, 8).double()          # [workgroup][wave][slot]
t = t[t[:, :, 5].sum(1) > 0]
for role, sl in (("group A waves 0-3 (logits, input gradient)", slice(0, 4)), ("group A waves 4-7 (input gradient)", slice(4, 8)),
                 ("group B waves 8-11 (copies incl. targets, staging, tap products, dW)", slice(8, 12)), ("group B waves 12-15", slice(12, 16))):
    m = t[:, sl, :].mean((0, 1))
    print(f"{role}: workgroups {t.shape[0]} ->", {n: f"{m[k].item():.0f}" for k, n in enumerate(names)})
