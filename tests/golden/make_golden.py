"""Generate tests/golden/*.npz by running the REFERENCE itself on CPU.

Run in the build container only (needs /root/reference):
    python tests/golden/make_golden.py

What is imported from the reference: midi_autoencoder/models.py (VanillaVAE)
and midi_autoencoder/train.py (train_one_epoch).  train.py imports torchvision
(absent here) through datasets.py / data_transformations.py, so four empty
stub modules are placed in sys.modules first; torch.cuda.Event is replaced by a
no-op because train.py:632-633 records a CUDA event unconditionally (SURVEY.md
8c).  torch.randn_like is patched to return the explicit eps draw (H5).

Inputs, weights and eps are regenerated on both sides from the counter-based
generator in oracle/vae_oracle.py, so the fixtures hold OUTPUTS only.
"""
import contextlib
import io
import os
import sys
import types
from argparse import Namespace

import numpy as np
import torch

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF = "/root/reference/midi_autoencoder"

from oracle import vae_oracle as vo  # noqa: E402

CASES = [
    # name, H, L, B, steps, total_steps, kld_weight, generalised, seed
    ("R_b4_k1", 32, 16, 4, 3, 10, 1.0, False, 1),
    ("R_b32_k1", 32, 16, 32, 20, 200, 1.0, False, 2),
    ("R_b32_k4", 32, 16, 32, 2, 10, 4.0, False, 3),
    ("R_b32_k16", 32, 16, 32, 2, 10, 16.0, False, 4),
    ("R_b256_k1", 32, 16, 256, 2, 10, 1.0, False, 5),
    ("G_h64_l16_b4", 64, 16, 4, 2, 10, 1.0, True, 6),
    ("G_h64_l64_b8", 64, 64, 8, 2, 10, 1.0, True, 7),
    ("G_h128_l16_b2", 128, 16, 2, 1, 10, 1.0, True, 8),
    ("G_h128_l128_b2", 128, 128, 2, 1, 10, 4.0, True, 9),
    # round 2: the remaining BASELINE.json shapes (configs[2]: 256x256 latent 64; configs[4]: 128x128 latent 128, beta 1/16)
    # and the reference's own default latent size 10 (train.py:875-877), which is not a multiple of 4
    ("G_h256_l64_b2", 256, 64, 2, 1, 10, 1.0, True, 10),
    ("G_h128_l128_b2_k1", 128, 128, 2, 1, 10, 1.0, True, 11),
    ("G_h128_l128_b2_k16", 128, 128, 2, 1, 10, 16.0, True, 12),
    ("R_l10_b8", 32, 10, 8, 2, 10, 1.0, False, 13),
]
ROUND1 = 9   # the first nine cases are the round-1 fixtures (not regenerated unless --all)


def import_reference():
    for name in ["torchvision", "torchvision.datasets", "torchvision.transforms", "torchvision.transforms.v2"]:
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.modules["torchvision.datasets"].ImageFolder = object
    sys.modules["torchvision.transforms.v2"].Transform = object
    sys.modules["torchvision"].datasets = sys.modules["torchvision.datasets"]
    sys.modules["torchvision"].transforms = sys.modules["torchvision.transforms"]
    sys.modules["torchvision.transforms"].v2 = sys.modules["torchvision.transforms.v2"]
    sys.path.insert(0, REF)
    import models  # noqa
    import train  # noqa

    class _NoEvent:
        def __init__(self, *a, **k):
            pass

        def record(self):
            pass

        def elapsed_time(self, other):
            return 0.0

    torch.cuda.Event = _NoEvent
    return models, train


def build_model(models, H, L, generalised, kld_weight, params, dtype):
    with contextlib.redirect_stdout(io.StringIO()):
        if not generalised:
            m = models.VanillaVAE(1, L, input_dim=H, kld_weight=kld_weight)
        else:
            s = H // 16

            class GeneralisedVAE(models.VanillaVAE):
                """SURVEY.md 8c 'generalised oracle': flattened_size = 256*(H/16)^2.
                NOT reference behaviour (models.py:33,166 hard-wire 32x32)."""

                def __init__(self):
                    super().__init__(1, L, input_dim=H, kld_weight=kld_weight)
                    self.last_conv_size = s * s
                    self.flattened_size = 256 * s * s
                    self.fc_mu = torch.nn.Linear(self.flattened_size, L)
                    self.fc_var = torch.nn.Linear(self.flattened_size, L)
                    self.decoder_input = torch.nn.Linear(L, self.flattened_size)

                def decode(self, z):
                    x = self.decoder_input(z).view(-1, 256, s, s)
                    return self.final_layer(self.decoder(x))

            m = GeneralisedVAE()
    m = m.to(dtype)
    sd = m.state_dict()
    for k, v in params.items():
        assert tuple(sd[k].shape) == v.shape, (k, sd[k].shape, v.shape)
        sd[k] = torch.from_numpy(v).to(dtype)
    m.load_state_dict(sd)
    return m


def run_case(models, train, case, dtype, checkpoint_to=None):
    name, H, L, B, steps, total_steps, kw, gen, seed = case
    params = vo.init_params(L, H, seed, gen, np.float64)
    model = build_model(models, H, L, gen, kw, params, dtype)
    # optimiser + scheduler exactly as train.py:201-238
    lr = 0.01 * B * 1 / 128
    groups = [
        {"params": model.encoder.parameters(), "lr": lr * 1.0, "name": "encoder"},
        {"params": model.decoder.parameters(), "lr": lr * 1.0, "name": "decoder"},
    ]
    optimizer = torch.optim.AdamW(groups, lr=lr, weight_decay=0.0)
    scheduler = torch.optim.lr_scheduler.OneCycleLR(
        optimizer, [g["lr"] for g in optimizer.param_groups], epochs=1, steps_per_epoch=total_steps)
    xs = [torch.from_numpy(vo.synth_pianoroll(B, H, seed * 1000 + s)).to(dtype) for s in range(steps)]
    epss = [torch.from_numpy(vo.counter_normal(B * L, seed * 1000 + s, 5).reshape(B, L)).to(dtype) for s in range(steps)]
    out = {}
    queue = list(epss)
    orig = torch.randn_like
    torch.randn_like = lambda t, **k: queue.pop(0).to(t.dtype)
    try:
        # step 0 by hand to capture intermediates (same op order as train.py:634-659)
        model.train()
        o = model.forward(xs[0])
        optimizer.zero_grad()
        lo = model.loss(o)
        lo["loss"].backward()
        out["mu"] = o["encoded"]["mu"].detach().numpy()
        out["log_var"] = o["encoded"]["log_var"].detach().numpy()
        out["latents"] = o["latents"].detach().numpy()
        out["pre_latents_sum"] = np.array(o["encoded"]["pre_latents"].detach().double().sum().item())
        xhat = o["output"].detach().double().numpy()
        out["output_sum"] = np.array(xhat.sum())
        out["output_l2"] = np.array(np.sqrt((xhat ** 2).sum()))
        idx = (vo.counter_uniform(64, seed, 99) * xhat.size).astype(np.int64)
        out["output_idx"] = idx
        out["output_samples"] = xhat.reshape(-1)[idx]
        for n, p in model.named_parameters():
            g = p.grad.detach().double().numpy().reshape(-1)
            out["gradnorm/" + n] = np.array(np.sqrt((g ** 2).sum()))
            gi = (vo.counter_uniform(16, seed, 123) * g.size).astype(np.int64)
            out["gradsamp/" + n] = g[gi]
        losses = [[lo["loss"].item(), lo["reconstruction_loss"].item(), lo["kld_loss"].item()]]
        lrs = [[optimizer.param_groups[0]["lr"], optimizer.param_groups[0]["betas"][0]]]
        optimizer.step()
        scheduler.step()
        # remaining steps through the reference's own train_one_epoch
        if steps > 1:
            cfg = Namespace(log_wandb=False, print_interval=1000, log_interval=1000,
                            freeze_encoder=False, world_size=1, global_rank=0)
            rec = []

            def crit(o_):
                lrs.append([optimizer.param_groups[0]["lr"], optimizer.param_groups[0]["betas"][0]])
                r = model.loss(o_)
                rec.append([r["loss"].item(), r["reconstruction_loss"].item(), r["kld_loss"].item()])
                return r

            loader = [(x, torch.zeros(B, dtype=torch.long)) for x in xs[1:]]
            with contextlib.redirect_stdout(io.StringIO()):
                res, total_step, n_seen = train.train_one_epoch(
                    cfg, model, optimizer, scheduler, crit, loader, device="cpu", epoch=2,
                    total_step=1, n_samples_seen=B)
            losses += rec
            out["epoch_loss"] = np.array(res["loss"])
            out["total_step"] = np.array(total_step)
            out["n_samples_seen"] = np.array(n_seen)
        if checkpoint_to is not None:
            # N2: a checkpoint written by the reference's own utils.safe_save_model (utils.py:311-351) with the modules
            # train.py:444-460 passes, then two more steps of the reference's loop from that state (the trajectory a
            # resumed run must continue)
            import utils as ref_utils
            cfg_ck = Namespace(checkpoint_path=checkpoint_to, global_rank=1, lr=lr, batch_size=B)
            with contextlib.redirect_stdout(io.StringIO()):
                ref_utils.safe_save_model({"encoder": model.encoder, "decoder": model.decoder, "optimizer": optimizer,
                                           "scheduler": scheduler}, checkpoint_to, config=cfg_ck, epoch=2, total_step=steps,
                                          n_samples_seen=steps * B, best_epoch=0)
            nxt = []
            xs2 = [torch.from_numpy(vo.synth_pianoroll(B, H, seed * 1000 + s)).to(dtype) for s in range(steps, steps + 2)]
            queue.extend(torch.from_numpy(vo.counter_normal(B * L, seed * 1000 + s, 5).reshape(B, L)).to(dtype) for s in range(steps, steps + 2))

            def crit2(o_):
                r = model.loss(o_)
                nxt.append([r["loss"].item(), r["reconstruction_loss"].item(), r["kld_loss"].item()])
                return r

            cfg = Namespace(log_wandb=False, print_interval=1000, log_interval=1000, freeze_encoder=False, world_size=1, global_rank=0)
            with contextlib.redirect_stdout(io.StringIO()):
                train.train_one_epoch(cfg, model, optimizer, scheduler, crit2, [(x, torch.zeros(B, dtype=torch.long)) for x in xs2],
                                      device="cpu", epoch=3, total_step=steps, n_samples_seen=steps * B)
            out["resumed_losses"] = np.array(nxt, dtype=np.float64)
    finally:
        torch.randn_like = orig
    out["losses"] = np.array(losses, dtype=np.float64)
    out["lr_beta1"] = np.array(lrs, dtype=np.float64)
    for n, p in model.named_parameters():
        v = p.detach().double().numpy()
        out["param_sum/" + n] = np.array(v.sum())
        out["param_l2/" + n] = np.array(np.sqrt((v ** 2).sum()))
    for n, b in model.named_buffers():
        out["buf/" + n] = b.detach().double().numpy()
    return out


def evaluate_fixture(models):
    """N3: the reference's evaluation.evaluate (evaluation.py:12-113) on a two-batch loader whose dataset is one sample
    shorter than the batches (the DistributedSampler-padding trim at :88-95), eval mode, fresh running statistics."""
    import evaluation as ref_eval
    H, L, B, seed, n_samples = 32, 16, 4, 21, 7
    params = vo.init_params(L, H, seed, False, np.float64)
    model = build_model(models, H, L, False, 1.0, params, torch.float32)

    class Loader(list):
        pass

    loader = Loader((torch.from_numpy(vo.synth_pianoroll(B, H, seed * 1000 + i)), torch.zeros(B, dtype=torch.long)) for i in range(2))
    loader.dataset = range(n_samples)
    queue = [torch.from_numpy(vo.counter_normal(B * L, seed * 1000 + i, 5).reshape(B, L)).float() for i in range(2)]
    orig = torch.randn_like
    torch.randn_like = lambda t, **k: queue.pop(0).to(t.dtype)
    try:
        with contextlib.redirect_stdout(io.StringIO()):
            res = ref_eval.evaluate(loader, model, "cpu", verbosity=0)
    finally:
        torch.randn_like = orig
    return {k: np.array(float(v)) for k, v in res.items()}


def bce_fixture():
    """ATen F.binary_cross_entropy edge behaviour (log clamp -100, grad clamp 1e-12)."""
    x = torch.tensor([0.0, 1.0, 0.0, 1.0, 1e-30, 1 - 1e-7, 0.5, 0.25, 1e-13, 0.999999], dtype=torch.float32, requires_grad=True)
    t = torch.tensor([1.0, 0.0, 0.0, 1.0, 1.0, 0.0, 0.3, 1.0, 0.5, 0.0], dtype=torch.float32)
    per = torch.nn.functional.binary_cross_entropy(x, t, reduction="none")
    per.sum().backward()
    return {"x": x.detach().numpy(), "t": t.numpy(), "per_elem": per.detach().numpy(), "grad": x.grad.numpy()}


def onecycle_fixture():
    out = {}
    for total in (10, 200, 1001):
        p = torch.nn.Parameter(torch.zeros(1))
        opt = torch.optim.AdamW([{"params": [p], "lr": 0.02}], lr=0.02)
        sch = torch.optim.lr_scheduler.OneCycleLR(opt, [0.02], epochs=1, steps_per_epoch=total)
        tr = []
        for _ in range(total):
            tr.append([opt.param_groups[0]["lr"], opt.param_groups[0]["betas"][0]])
            opt.step()
            sch.step()
        out[f"trace_{total}"] = np.array(tr)
    return out


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    models, train = import_reference()
    cases = CASES if "--all" in sys.argv else CASES[ROUND1:]
    for case in cases:
        for dtype, tag in ((torch.float32, "f32"), (torch.float64, "f64")):
            out = run_case(models, train, case, dtype)
            path = os.path.join(HERE, f"{case[0]}_{tag}.npz")
            np.savez_compressed(path, **out)
            print(case[0], tag, "loss", out["losses"][0], os.path.getsize(path), "bytes")
    # N2: reference-written checkpoint (f32, after the 3 steps of R_b4_k1) + the losses of the two steps that follow it
    ck = os.path.join(HERE, "ckpt_R_b4_k1_f32.pt")
    out = run_case(models, train, CASES[0], torch.float32, checkpoint_to=ck)
    np.savez_compressed(os.path.join(HERE, "ckpt_R_b4_k1_f32_next.npz"), resumed_losses=out["resumed_losses"], losses=out["losses"])
    print("checkpoint", os.path.getsize(ck), "bytes; resumed losses", out["resumed_losses"][:, 0])
    np.savez_compressed(os.path.join(HERE, "evaluate_R.npz"), **evaluate_fixture(models))
    if "--all" in sys.argv:
        np.savez_compressed(os.path.join(HERE, "bce_edges.npz"), **bce_fixture())
        np.savez_compressed(os.path.join(HERE, "onecycle.npz"), **onecycle_fixture())


if __name__ == "__main__":
    main()
