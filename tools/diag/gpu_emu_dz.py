"""Diagnostic (GPU box): the stored dz tensors of a 16-bit mode against the storage-emulating oracle's, layer by layer in backward
order: relative L2, fraction of elements off by more than 2 storage ulps, fraction whose ratio says "other LeakyReLU slope".
python tools/diag/gpu_emu_dz.py H L B dtype [gen]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from oracle import vae_oracle as vo
from torch_vae_amd import _lib
from tests.util import make_model, perturbed_params
H, L, B, dtype = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
gen = (sys.argv[5] == "1") if len(sys.argv) > 5 else H != 32
p = perturbed_params(L, H, 17, gen)
x = vo.synth_pianoroll(B, H, 3); eps = vo.counter_normal(B * L, 3, 5).reshape(B, L)
m = make_model(H, L, gen, dtype, p)
m.fused_forward_backward(torch.from_numpy(x).cuda(), eps=torch.from_numpy(eps).float().cuda())
ce = vo.forward(p, x.astype(np.float64), eps, None, train=True, storage=dtype); ge = vo.backward(p, ce)
names = ["encoder.0", "encoder.1", "encoder.2", "encoder.3", "decoder.0", "decoder.1", "decoder.2", "final_layer"]
gs = vo.f16_grad_scale(B, H) if dtype == "f16" else 1.0
ulp = {"bf16": 2.0 ** -8, "f16": 2.0 ** -11}[dtype]
for i in (7, 6, 5, 4, 3, 2, 1, 0):
    want = ge[names[i] + ".dz"]
    n = want.size
    got = torch.empty(n, device="cuda")
    _lib.check(_lib.lib().vae_debug_tensor(m._ctx.handle, 8 + i, got.data_ptr(), n, torch.cuda.current_stream().cuda_stream), "dbg")
    got = got.cpu().numpy().reshape(want.shape).astype(np.float64)
    for scale in (1.0, gs):
        if scale != 1.0 or gs == 1.0:
            pass
    # the kernels may hold dz multiplied by the gradient scale: pick the better fit
    r1 = np.linalg.norm(got - want) / np.linalg.norm(want); r2 = np.linalg.norm(got / gs - want) / np.linalg.norm(want)
    if r2 < r1: got = got / gs
    rel = np.linalg.norm(got - want) / np.linalg.norm(want)
    big = np.abs(got - want) > 4 * ulp * np.maximum(np.abs(want), 1e-30)
    nz = np.abs(want) > 0
    ratio = np.where(nz, got / np.where(nz, want, 1), 1.0)
    flip = ((ratio > 50) & (ratio < 200)) | ((ratio > 0.005) & (ratio < 0.02))
    # how the mismatching elements relate: error relative to the layer's rms
    rms = np.sqrt((want ** 2).mean())
    err = np.abs(got - want)
    print(f"{names[i]:14s} rel_l2 {rel:.4f}  >4ulp {big.mean():.2e}  slope-flip-like {flip.mean():.2e}  max err/rms {err.max() / rms:.2f}  "
          f"err mass in flips {float((err[flip] ** 2).sum() / max((err ** 2).sum(), 1e-300)):.2f}", flush=True)

# ---- the first layer of the backward (final_layer): what do the slope-flip-like elements look like?
i = 7
want = ge["final_layer.dz"]; n = want.size
got = torch.empty(n, device="cuda"); yk = torch.empty(n, device="cuda")
_lib.check(_lib.lib().vae_debug_tensor(m._ctx.handle, 8 + i, got.data_ptr(), n, torch.cuda.current_stream().cuda_stream), "dbg")
_lib.check(_lib.lib().vae_debug_tensor(m._ctx.handle, i, yk.data_ptr(), n, torch.cuda.current_stream().cuda_stream), "dbg")
got = got.cpu().numpy().reshape(want.shape).astype(np.float64); yk = yk.cpu().numpy().reshape(want.shape).astype(np.float64)
if np.linalg.norm(got / gs - want) < np.linalg.norm(got - want): got = got / gs
yo, zo = ce["final_layer.y"], ce["final_layer.z"]
print("stored y7: kernel vs oracle differing fraction", float((yk != yo).mean()), " rel_l2", float(np.linalg.norm(yk - yo) / np.linalg.norm(yo)))
nzm = np.abs(want) > 0
ratio = np.where(nzm, got / np.where(nzm, want, 1), 1.0)
flip = ((ratio > 50) & (ratio < 200)) | ((ratio > 0.005) & (ratio < 0.02))
idx = np.argwhere(flip)
print("flip-like elements:", len(idx), "of", n, "; of them with kernel y != oracle y:", int((yk[flip] != yo[flip]).sum()),
      "; |z_oracle| quantiles", np.quantile(np.abs(zo[flip]), [0.1, 0.5, 0.9]), " all |z| median", float(np.median(np.abs(zo))))
for b, c_, yy, xx in idx[:12]:
    print(f"  ch {c_:2d} y_oracle {yo[b, c_, yy, xx]:+.6f} y_kernel {yk[b, c_, yy, xx]:+.6f} z_oracle {zo[b, c_, yy, xx]:+.3e} dz want {want[b, c_, yy, xx]:+.3e} got {got[b, c_, yy, xx]:+.3e}")
