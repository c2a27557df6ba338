"""Diagnostic (GPU box): the row-streaming final_layer.0 forward kernel against the tiled one: y7 (debug tensor 7), xhat, ELBO,
gradients - same arithmetic up to the order of the f32 accumulation."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from torch_vae_amd import _lib
from torch_vae_amd.models import VanillaVAE
from torch_vae_amd.train import SyntheticPianorollLoader
for H, L, B, dt in ((128, 16, 3, "bf16"), (128, 16, 5, "f16"), (128, 16, 2, "bf16"), (128, 16, 256, "bf16")):
    res = []
    for stream in (0, 1):
        torch.manual_seed(1)
        model = VanillaVAE(1, L, H, generalised=True, compute_dtype=dt, max_batch=B).cuda()
        x = SyntheticPianorollLoader(B, H, 1, device="cuda").batch(0)[0]
        eps = torch.randn(B, L, device="cuda", generator=torch.Generator(device="cuda").manual_seed(3))
        _lib.check(_lib.lib().vae_set_option(model._context(B).handle, b"use_upf_stream", stream), "set")
        out3, xhat = model.fused_forward_backward(x, eps=eps)
        n = B * 32 * H * H
        y7 = torch.empty(n, device="cuda")
        _lib.check(_lib.lib().vae_debug_tensor(model._ctx.handle, 7, y7.data_ptr(), n, torch.cuda.current_stream().cuda_stream), "dbg")
        torch.cuda.synchronize()
        res.append((out3.clone(), xhat.clone(), y7, model.flat_grads().clone(), model._bnflat.clone()))
    (o0, x0, y0, g0, b0), (o1, x1, y1, g1, b1) = res
    ndiff = int((y0 != y1).sum()); rel = float((y0 - y1).norm() / y0.norm())
    print(H, L, B, dt, f"y7 differing {ndiff}/{y0.numel()} rel {rel:.2e} | xhat max abs {float((x0-x1).abs().max()):.2e} | ELBO rel {float(((o0-o1)/o0).abs().max()):.2e}"
          f" | grads rel {float((g0-g1).norm()/g0.norm()):.2e} | bn rel {float((b0-b1).norm()/b0.norm()):.2e}", flush=True)
