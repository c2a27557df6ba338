"""Diagnostic (GPU box): d0 (decoder_input output, id 16) and dd0 (17) under two option sets, against the oracle."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from oracle import vae_oracle as vo
from tests.util import make_model, perturbed_params, rel_l2
from torch_vae_amd import _lib
H, L, B, dtype = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
gen = H != 32
p = perturbed_params(L, H, 17, gen)
x = vo.synth_pianoroll(B, H, 3); eps = vo.counter_normal(B * L, 3, 5).reshape(B, L)
c = vo.forward(p, x.astype(np.float64), eps, None, train=True)
d0_ref = (c["zlat"] @ p["decoder_input.weight"].T + p["decoder_input.bias"])     # [B, F_ref]  (NCHW flatten: c*s2 + pix)
outs = []
for opts in sys.argv[5:7]:
    m = make_model(H, L, gen, dtype, p); m._context(B)
    for kv in filter(None, opts.split(",")):
        k, v = kv.split("="); assert _lib.lib().vae_set_option(m._ctx.handle, k.encode(), int(v)) == 0
    m.fused_forward_backward(torch.from_numpy(x).cuda(), eps=torch.from_numpy(eps).float().cuda())
    n = B * m.flattened_size
    buf = torch.empty(n, device="cuda")
    _lib.check(_lib.lib().vae_debug_tensor(m._ctx.handle, 16, buf.data_ptr(), n, torch.cuda.current_stream().cuda_stream), "dbg")
    d0 = buf.cpu().numpy().reshape(B, -1)        # debug tensors come back NCHW
    outs.append(d0)
    err = np.abs(d0 - d0_ref)
    print(opts, "d0 rel_l2 vs oracle", rel_l2(d0, d0_ref), "max abs err", err.max(), "at", np.unravel_index(err.argmax(), err.shape), "n bad(>1e-4)", int((err > 1e-4).sum()))
    bad = np.argwhere(err > 1e-4)
    if len(bad):
        f = bad[:, 1]; s2 = m.flattened_size // 256
        print("  bad channels", sorted(set((f // s2).tolist()))[:40], "bad pixels", sorted(set((f % s2).tolist()))[:70], "rows", sorted(set(bad[:, 0].tolist())))
