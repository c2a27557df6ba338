"""Diagnostic (GPU box): a kernel option on / off - the stored conv output of one BatchNorm layer (debug tensor `which`), ELBO, gradients.
python tools/diag/gpu_ab_layer.py use_dnf_stream 1"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from torch_vae_amd import _lib
from torch_vae_amd.models import VanillaVAE
from torch_vae_amd.train import SyntheticPianorollLoader
opt, which = sys.argv[1].encode(), int(sys.argv[2])
C = [32, 64, 128, 256, 128, 64, 32, 32][which]
for H, L, B, dt in ((128, 16, 3, "bf16"), (128, 16, 5, "f16"), (128, 16, 40, "bf16"), (128, 16, 256, "bf16")):
    S = [H // 2, H // 4, H // 8, H // 16, H // 8, H // 4, H // 2, H][which]
    res = []
    for use in (0, 1):
        torch.manual_seed(1)
        model = VanillaVAE(1, L, H, generalised=True, compute_dtype=dt, max_batch=B).cuda()
        x = SyntheticPianorollLoader(B, H, 1, device="cuda").batch(0)[0]
        eps = torch.randn(B, L, device="cuda", generator=torch.Generator(device="cuda").manual_seed(3))
        _lib.check(_lib.lib().vae_set_option(model._context(B).handle, opt, use), "set")
        out3, xhat = model.fused_forward_backward(x, eps=eps)
        n = B * C * S * S
        y = torch.empty(n, device="cuda")
        _lib.check(_lib.lib().vae_debug_tensor(model._ctx.handle, which, y.data_ptr(), n, torch.cuda.current_stream().cuda_stream), "dbg")
        torch.cuda.synchronize()
        res.append((out3.clone(), y, model.flat_grads().clone(), model._bnflat.clone()))
    (o0, y0, g0, b0), (o1, y1, g1, b1) = res
    nd = int((y0 != y1).sum())
    print(H, L, B, dt, f"y{which} differing {nd}/{y0.numel()} rel {float((y0 - y1).norm() / y0.norm()):.2e} max abs {float((y0 - y1).abs().max()):.2e}"
          f" | ELBO rel {float(((o0 - o1) / o0).abs().max()):.2e} | grads rel {float((g0 - g1).norm() / g0.norm()):.2e} | bn rel {float((b0 - b1).norm() / b0.norm()):.2e}", flush=True)
