"""Diagnostic (GPU box): per-wave cycle shares of the workgroup-specialised deep-layer kernels (conv_deep.cuh).
Needs the stamps build:  make -C torch_vae_amd/csrc STAMPS=1 OUT=../lib/libvae_step_stamps.so OBJD=../lib/obj_stamps
run as:  VAE_STEP_LIB=torch_vae_amd/lib/libvae_step_stamps.so python tools/diag/gpu_stamps_deep.py"""
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
names = ["prologue(after setup)", "fill", "work", "step_barrier", "epilogue", "total", "tile_wait", "setup", "dma_issue", "dma_wait", "pro:dma0", "pro:coef+acc", "pro:orel"]
for tag, epi in [("encoder.3", 0), ("encoder.2", 0), ("decoder.0", 2), ("decoder.1", 1)]:
    buf = torch.zeros(256 * 8 * 16, dtype=torch.int64, device="cuda")
    Lb.vae_debug_stamps(model._ctx.handle, tag.encode(), epi, buf.data_ptr())
    model.fused_forward_backward(x); torch.cuda.synchronize()
    Lb.vae_debug_stamps(model._ctx.handle, b"", 0, None)
    t = buf.view(-1, 8, 16).double()          # [workgroup][wave][slot]
    t = t[t[:, :, 5].sum(1) > 0]
    for role, sl in (("consumers", slice(0, 4)), ("cons w0-1", slice(0, 2)), ("cons w2-3", slice(2, 4)), ("producers", slice(4, 8))):
        m = t[:, sl, :].mean((0, 1))
        print(f"{tag} epi={epi} {role}: workgroups {t.shape[0]} ->", {n: f"{m[k].item():.0f}" for k, n in enumerate(names)})
