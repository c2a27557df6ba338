"""Knob sweep on the GPU box (not a pytest file): times the step with different persistent-grid sizes."""
import sys, os, time, itertools, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from argparse import Namespace
from torch_vae_amd import _lib
from torch_vae_amd.models import VanillaVAE
from torch_vae_amd.train import SyntheticPianorollLoader, build_optimizer
H, L, B = 128, 16, 256
model = VanillaVAE(1, L, H, generalised=True, compute_dtype="bf16", max_batch=B).cuda()
cfg = Namespace(batch_size_per_gpu=B, world_size=1, lr_relative=0.01, weight_decay=0.0, optimizer="AdamW", scheduler="OneCycle", epochs=1, freeze_encoder=False)
opt, sched = build_optimizer(cfg, model, steps_per_epoch=100000)
x = SyntheticPianorollLoader(B, H, 1, device="cuda").batch(0)[0]
model.fused_forward_backward(x); opt.step()
knobs = json.loads(sys.argv[1]) if len(sys.argv) > 1 else {"knob_up_per_cu": [3, 4], "knob_convout_grid": [1024, 2048]}
names = list(knobs)
for combo in itertools.product(*[knobs[n] for n in names]):
    for n, v in zip(names, combo):
        assert _lib.lib().vae_set_option(model._ctx.handle, n.encode(), v) == 0
    for _ in range(3):
        model.fused_forward_backward(x); opt.step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20):
        model.fused_forward_backward(x); opt.step()
    torch.cuda.synchronize()
    print(dict(zip(names, combo)), f"{(time.perf_counter()-t0)/20*1e3:.3f} ms/step", flush=True)
