"""Diagnostic (GPU box): the producer / consumer weight-gradient kernel against the 8-wave kernel: gradients must be bit-identical."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from torch_vae_amd import _lib
from torch_vae_amd.models import VanillaVAE
from torch_vae_amd.train import SyntheticPianorollLoader
for H, L, B, dt in ((32, 16, 5, "bf16"), (64, 16, 7, "f16"), (128, 16, 3, "bf16"), (32, 8, 33, "f16"), (128, 16, 256, "bf16")):
    torch.manual_seed(1)
    res = []
    for split in (0, 1):
        torch.manual_seed(1)
        model = VanillaVAE(1, L, H, generalised=True, compute_dtype=dt, max_batch=B).cuda()
        x = SyntheticPianorollLoader(B, H, 1, device="cuda").batch(0)[0]
        eps = torch.randn(B, L, device="cuda", generator=torch.Generator(device="cuda").manual_seed(3))
        _lib.check(_lib.lib().vae_set_option(model._context(B).handle, b"use_wgrad_split", split), "set")
        out3, _ = model.fused_forward_backward(x, eps=eps)
        torch.cuda.synchronize()
        res.append((out3.clone(), model.flat_grads().clone()))
    same = torch.equal(res[0][1], res[1][1])
    d = (res[0][1] - res[1][1]).abs().max().item()
    print(H, L, B, dt, "bit-identical" if same else f"DIFFERENT max abs {d:.3e}", flush=True)
