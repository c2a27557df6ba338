"""Shared helpers for the parity tests."""
import numpy as np

from oracle import vae_oracle as vo

CASES = {
    # name: H, L, B, steps, total_steps, kld_weight, generalised, seed   (the table tests/golden/make_golden.py generates)
    "R_b4_k1": (32, 16, 4, 3, 10, 1.0, False, 1),
    "R_b32_k1": (32, 16, 32, 20, 200, 1.0, False, 2),
    "R_b32_k4": (32, 16, 32, 2, 10, 4.0, False, 3),
    "R_b32_k16": (32, 16, 32, 2, 10, 16.0, False, 4),
    "R_b256_k1": (32, 16, 256, 2, 10, 1.0, False, 5),
    "G_h64_l16_b4": (64, 16, 4, 2, 10, 1.0, True, 6),
    "G_h64_l64_b8": (64, 64, 8, 2, 10, 1.0, True, 7),
    "G_h128_l16_b2": (128, 16, 2, 1, 10, 1.0, True, 8),
    "G_h128_l128_b2": (128, 128, 2, 1, 10, 4.0, True, 9),
    # BASELINE.json configs[2] / configs[4] shapes and the reference's default latent size (round 2)
    "G_h256_l64_b2": (256, 64, 2, 1, 10, 1.0, True, 10),
    "G_h128_l128_b2_k1": (128, 128, 2, 1, 10, 1.0, True, 11),
    "G_h128_l128_b2_k16": (128, 128, 2, 1, 10, 16.0, True, 12),
    "R_l10_b8": (32, 10, 8, 2, 10, 1.0, False, 13),
}

# conv biases that feed a train-mode BatchNorm: analytically zero gradient
PRE_BN_BIAS = tuple([f"encoder.{i}.0.bias" for i in range(4)] + [f"decoder.{i}.0.bias" for i in range(3)]
                    + ["final_layer.0.bias"])


def rel_l2(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return float(np.sqrt(((a - b) ** 2).sum()) / max(np.sqrt((b ** 2).sum()), 1e-30))


def perturbed_params(L, H, seed, gen):
    """Reference-distribution init with non-trivial BN affine parameters and biases."""
    p = vo.init_params(L, H, seed, gen)
    rng = np.random.default_rng(seed)
    for k in p:
        if k.endswith(".1.weight"):
            p[k] = 1 + 0.2 * rng.standard_normal(p[k].shape)
        if k.endswith(".1.bias") or k.endswith(".0.bias") or k.endswith("3.bias"):
            p[k] = 0.1 * rng.standard_normal(p[k].shape)
    return p


def load_params(model, p):
    import torch
    sd = model.state_dict()
    for k, v in p.items():
        sd[k] = torch.from_numpy(np.asarray(v)).float()
    model.load_state_dict(sd)


def make_model(H, L, gen, dtype, p=None, kld_weight=1.0, device="cuda"):
    from torch_vae_amd.models import VanillaVAE
    m = VanillaVAE(1, L, H, kld_weight=kld_weight, generalised=gen, compute_dtype=dtype).to(device)
    if p is not None:
        load_params(m, p)
    return m


def flat_grad_dict(model):
    from torch_vae_amd import _lib
    g = model.flat_grads().detach().cpu().numpy()
    return {n: g[model._offs[i]:model._offs[i] + model._sizes[i]].copy() for i, n in enumerate(_lib.PARAM_NAMES)}
