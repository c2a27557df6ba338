"""Diagnostic (GPU box): the eight-wave variant of the 64x32-channel weight-gradient tile against the four-wave one
(same results expected: a wave owns whole taps, the K order does not change), then step time with either."""
import json, os, subprocess, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from oracle import vae_oracle as vo
from torch_vae_amd import _lib
from util import make_model, perturbed_params

H, L, B = 128, 16, 256
p = perturbed_params(L, H, 5, True)
x = torch.from_numpy(vo.synth_pianoroll(B, H, 3)).cuda()
eps = torch.from_numpy(vo.counter_normal(B * L, 3, 5).reshape(B, L)).float().cuda()
gs = []
for k in (0, 1):
    m = make_model(H, L, True, "bf16", p)
    _lib.check(_lib.lib().vae_set_option(m._context(B).handle, b"knob_wgrad_mid8", k), "set")
    m.fused_forward_backward(x, eps=eps)
    gs.append(m.flat_grads().detach().clone())
print("identical gradients:", bool(torch.equal(gs[0], gs[1])), "max abs diff", float((gs[0] - gs[1]).abs().max()))
_lib.check(_lib.lib().vae_set_option(m._context(B).handle, b"knob_wgrad_mid8", 0), "set")
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for r in range(3):
    for k in (0, 1):
        o = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "40", "--warmup", "5", "--no-cpu-baseline",
                            "--set", f"knob_wgrad_mid8={k}"], capture_output=True, text=True, cwd=root).stdout
        print("mid8 =", k, json.loads(o)["ms_per_step"], flush=True)
