"""Shared helpers for the parity tests."""
import numpy as np

from oracle import vae_oracle as vo

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
