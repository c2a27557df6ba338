"""CPU oracle for the VanillaVAE training step (TEST INFRASTRUCTURE ONLY).

This is a numpy restatement of the reference's per-step hot path.  It is the
checker for the HIP path; nothing in the product package may import it.  Only
tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it.

Parity status: PINNED against the reference itself.  tests/golden/*.npz were
generated in the build container by importing /root/reference/midi_autoencoder
(models.py, train.py) via tests/golden/make_golden.py; tests/test_oracle.py
checks this file against those fixtures.  The reference has no tests or golden
vectors of its own.

Reference lines followed (relative to /root/reference/midi_autoencoder):
  models.py:41-51   encoder blocks  Conv2d(k3,s2,p1) -> BatchNorm2d -> LeakyReLU
  models.py:55-59   fc_mu / fc_var / decoder_input Linear layers
  models.py:62-82   decoder ConvTranspose2d(k3,s2,p1,op1) blocks, final_layer
  models.py:107-145 encode      models.py:147-175 decode
  models.py:177-183 reparameterize   models.py:185-188 forward
  models.py:190-225 loss (BCE mean + kld_weight * KL, kld_loss sign flip)
  models.py:227-236 _init_weights (xavier on Conv2d of encoder/final_layer only)
  train.py:201-238  lr scaling, AdamW over encoder+decoder groups, OneCycleLR
  train.py:634-664  step order: forward, zero_grad, loss, backward, step, sched

Arithmetic conventions restated from PyTorch ATen:
BCE log clamp -100, BCE grad (x-t)/max(x(1-x),1e-12); BatchNorm biased batch
variance, eps 1e-5, running var unbiased, momentum 0.1; LeakyReLU slope 0.01;
AdamW decoupled decay, bias correction with the current beta1; OneCycleLR
two-phase cosine, div 25, final_div 1e4, beta1 0.95 <-> 0.85.

Tensors use the reference's NCHW layout; dtype is selectable (float64 for the
ground truth, float32 to mimic the reference arithmetic).
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field

import numpy as np

LEAKY_SLOPE = 0.01
BN_EPS = 1e-5
BN_MOMENTUM = 0.1
HIDDEN_DIMS = (32, 64, 128, 256)


# --------------------------------------------------------------------------
# counter-based generator (restated on both sides of every parity test)
# --------------------------------------------------------------------------
def _splitmix64(x: np.ndarray) -> np.ndarray:
    x = (x + np.uint64(0x9E3779B97F4A7C15)).astype(np.uint64)
    z = x.copy()
    z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    return z ^ (z >> np.uint64(31))


def counter_uniform(n: int, seed: int, stream: int) -> np.ndarray:
    """n doubles in [0,1): splitmix64(seed, stream, counter), top 53 bits."""
    with np.errstate(over="ignore"):
        base = _splitmix64(np.array([seed], dtype=np.uint64))
        base = _splitmix64(base ^ (np.uint64(stream) * np.uint64(0xD1342543DE82EF95)))
        ctr = np.arange(n, dtype=np.uint64) * np.uint64(0x2545F4914F6CDD1D)
        bits = _splitmix64(base + ctr)
    return (bits >> np.uint64(11)).astype(np.float64) * (1.0 / (1 << 53))


def counter_normal(n: int, seed: int, stream: int) -> np.ndarray:
    """Box-Muller on two uniform streams (deterministic, order-independent)."""
    u1 = counter_uniform(n, seed, 2 * stream + 1_000_003)
    u2 = counter_uniform(n, seed, 2 * stream + 1_000_004)
    return np.sqrt(-2.0 * np.log(1.0 - u1)) * np.cos(2.0 * np.pi * u2)


# --------------------------------------------------------------------------
# parameter census (names and shapes of models.VanillaVAE.state_dict())
# --------------------------------------------------------------------------
def param_shapes(latent_dim: int, img_size: int = 32, in_channels: int = 1,
                 generalised: bool = False) -> dict:
    """Shapes keyed by the reference's state_dict names (models.py:41-82).

    generalised=False is the reference-exact model (flattened_size = 4*256,
    models.py:33,36; only valid at 32x32).  generalised=True sets
    flattened_size = 256*(img_size/16)^2 ('generalised oracle').
    """
    hd = list(HIDDEN_DIMS)
    s = img_size // 16 if generalised else 2
    flat = hd[-1] * s * s
    shapes = {}
    cin = in_channels
    for i, h in enumerate(hd):
        shapes[f"encoder.{i}.0.weight"] = (h, cin, 3, 3)
        shapes[f"encoder.{i}.0.bias"] = (h,)
        shapes[f"encoder.{i}.1.weight"] = (h,)
        shapes[f"encoder.{i}.1.bias"] = (h,)
        cin = h
    shapes["fc_mu.weight"] = (latent_dim, flat)
    shapes["fc_mu.bias"] = (latent_dim,)
    shapes["fc_var.weight"] = (latent_dim, flat)
    shapes["fc_var.bias"] = (latent_dim,)
    shapes["decoder_input.weight"] = (flat, latent_dim)
    shapes["decoder_input.bias"] = (flat,)
    rev = hd[::-1]
    for i in range(len(rev) - 1):
        shapes[f"decoder.{i}.0.weight"] = (rev[i], rev[i + 1], 3, 3)
        shapes[f"decoder.{i}.0.bias"] = (rev[i + 1],)
        shapes[f"decoder.{i}.1.weight"] = (rev[i + 1],)
        shapes[f"decoder.{i}.1.bias"] = (rev[i + 1],)
    shapes["final_layer.0.weight"] = (rev[-1], rev[-1], 3, 3)
    shapes["final_layer.0.bias"] = (rev[-1],)
    shapes["final_layer.1.weight"] = (rev[-1],)
    shapes["final_layer.1.bias"] = (rev[-1],)
    shapes["final_layer.3.weight"] = (1, rev[-1], 3, 3)
    shapes["final_layer.3.bias"] = (1,)
    return shapes


def init_params(latent_dim: int, img_size: int = 32, seed: int = 0,
                generalised: bool = False, dtype=np.float64) -> dict:
    """Weights with the reference's init DISTRIBUTIONS from the counter RNG.

    models.py:227-236: xavier_uniform + zero bias for nn.Conv2d inside
    encoder/final_layer; BN weight 1 bias 0.  ConvTranspose2d and the three
    Linear layers keep torch defaults: kaiming_uniform(a=sqrt(5)) ->
    U(-1/sqrt(fan_in), 1/sqrt(fan_in)) for weight and bias (fan_in computed
    from weight.size(1)*receptive field, as torch does).
    """
    shapes = param_shapes(latent_dim, img_size, generalised=generalised)
    out = {}
    for stream, (name, shp) in enumerate(shapes.items()):
        n = int(np.prod(shp))
        u = counter_uniform(n, seed, stream).reshape(shp)
        is_bn = name.split(".")[-2] == "1" and len(shp) == 1 and not name.startswith("fc")
        if is_bn:
            val = np.ones(shp) if name.endswith("weight") else np.zeros(shp)
        elif name.startswith("encoder") or name == "final_layer.3.weight" or name == "final_layer.3.bias":
            if name.endswith("weight"):
                fan_in = shp[1] * 9
                fan_out = shp[0] * 9
                bound = math.sqrt(6.0 / (fan_in + fan_out))
                val = (2 * u - 1) * bound
            else:
                val = np.zeros(shp)
        else:
            wshape = shapes[name.rsplit(".", 1)[0] + ".weight"]
            fan_in = wshape[1] * (9 if len(wshape) == 4 else 1)
            bound = 1.0 / math.sqrt(fan_in)
            val = (2 * u - 1) * bound
        out[name] = np.ascontiguousarray(val, dtype=dtype)
    return out


def init_bn_state(dtype=np.float64) -> dict:
    st = {}
    names = [f"encoder.{i}.1" for i in range(4)] + [f"decoder.{i}.1" for i in range(3)] + ["final_layer.1"]
    chans = list(HIDDEN_DIMS) + [128, 64, 32] + [32]
    for n, c in zip(names, chans):
        st[n + ".running_mean"] = np.zeros(c, dtype=dtype)
        st[n + ".running_var"] = np.ones(c, dtype=dtype)
        st[n + ".num_batches_tracked"] = np.zeros((), dtype=np.int64)
    return st


def synth_pianoroll(batch: int, img_size: int, seed: int = 0, max_lines: int = 20) -> np.ndarray:
    """Seeded restatement of the line-image distribution (data_generators.py:45-77
    as called at :97-104): n_lines~U{1..20}, vertical w.p. 1/2, width drawn once
    from U{1..4} and then sticky, partial lines start~U[0,H), end~U[start,H),
    pixel 255 -> 1.0.  Returns float32 [B,1,H,H] in {0,1}."""
    H = img_size
    u = counter_uniform(batch * (1 + 4 * max_lines) + 1, seed, 777)
    k = 0
    width = 1 + int(u[k] * 4); k += 1
    x = np.zeros((batch, 1, H, H), dtype=np.float32)
    for b in range(batch):
        n_lines = 1 + int(u[k] * max_lines); k += 1
        for li in range(max_lines):
            vert = u[k] < 0.5
            pos = int(u[k + 1] * H)
            start = int(u[k + 2] * H)
            end = start + int(u[k + 3] * (H - start))
            k += 4
            if li >= n_lines:
                continue
            lo, hi = max(0, pos - width // 2), min(H, pos + width // 2 + 1)
            if vert:
                x[b, 0, start:end, lo:hi] = 1.0
            else:
                x[b, 0, lo:hi, start:end] = 1.0
    return x


# --------------------------------------------------------------------------
# layer primitives (NCHW)
# --------------------------------------------------------------------------
def _im2col(x, stride, pad):
    B, C, H, W = x.shape
    xp = np.pad(x, ((0, 0), (0, 0), (pad, pad), (pad, pad)))
    Ho = (H + 2 * pad - 3) // stride + 1
    Wo = (W + 2 * pad - 3) // stride + 1
    s = xp.strides
    cols = np.lib.stride_tricks.as_strided(
        xp, shape=(B, Ho, Wo, C, 3, 3),
        strides=(s[0], s[2] * stride, s[3] * stride, s[1], s[2], s[3]), writeable=False)
    return cols.reshape(B * Ho * Wo, C * 9), Ho, Wo


def conv_fwd(x, w, b, stride, pad=1):
    """nn.Conv2d(k=3): out[b,o,y,x] = bias[o] + sum w[o,c,ky,kx] x[b,c,s*y+ky-p,s*x+kx-p]."""
    B = x.shape[0]
    O = w.shape[0]
    cols, Ho, Wo = _im2col(x, stride, pad)
    out = cols @ w.reshape(O, -1).T
    if b is not None:
        out = out + b
    return np.ascontiguousarray(out.reshape(B, Ho, Wo, O).transpose(0, 3, 1, 2))


def conv_dgrad(dout, w, stride, in_hw, pad=1):
    """Gradient of conv_fwd w.r.t. its input (also ConvTranspose2d forward)."""
    B, O, Ho, Wo = dout.shape
    C = w.shape[1]
    H, W = in_hw
    d = dout.transpose(0, 2, 3, 1).reshape(B * Ho * Wo, O) @ w.reshape(O, C * 9)
    d = d.reshape(B, Ho, Wo, C, 3, 3)
    dxp = np.zeros((B, C, H + 2 * pad + 2, W + 2 * pad + 2), dtype=dout.dtype)
    for ky in range(3):
        for kx in range(3):
            dxp[:, :, ky:ky + stride * Ho:stride, kx:kx + stride * Wo:stride] += \
                d[:, :, :, :, ky, kx].transpose(0, 3, 1, 2)
    return np.ascontiguousarray(dxp[:, :, pad:pad + H, pad:pad + W])


def conv_wgrad(x, dout, stride, pad=1):
    B, O, Ho, Wo = dout.shape
    C = x.shape[1]
    cols, _, _ = _im2col(x, stride, pad)
    dw = dout.transpose(0, 2, 3, 1).reshape(-1, O).T @ cols
    return dw.reshape(O, C, 3, 3), dout.sum(axis=(0, 2, 3))


def convT_fwd(x, w, b):
    """nn.ConvTranspose2d(k3,s2,p1,output_padding=1); w is [Cin,Cout,3,3]."""
    B, C, H, W = x.shape
    out = conv_dgrad(x, w, 2, (2 * H, 2 * W))
    return out + b.reshape(1, -1, 1, 1)


def convT_bwd(x, w, dout):
    dx = conv_fwd(dout, w, None, 2)
    dw, _ = conv_wgrad(dout, x, 2)
    return dx, dw, dout.sum(axis=(0, 2, 3))


def bn_train_fwd(y, gamma, beta):
    """BatchNorm2d training forward: biased batch variance, eps 1e-5."""
    mean = y.mean(axis=(0, 2, 3))
    var = y.var(axis=(0, 2, 3))
    invstd = 1.0 / np.sqrt(var + y.dtype.type(BN_EPS))
    xhat = (y - mean.reshape(1, -1, 1, 1)) * invstd.reshape(1, -1, 1, 1)
    z = xhat * gamma.reshape(1, -1, 1, 1) + beta.reshape(1, -1, 1, 1)
    return z, (xhat, invstd, mean, var)


def bn_train_bwd(dz, gamma, cache):
    xhat, invstd, _, _ = cache
    n = dz.shape[0] * dz.shape[2] * dz.shape[3]
    dgamma = (dz * xhat).sum(axis=(0, 2, 3))
    dbeta = dz.sum(axis=(0, 2, 3))
    g = (gamma * invstd).reshape(1, -1, 1, 1)
    dy = g * (dz - dbeta.reshape(1, -1, 1, 1) / n - xhat * dgamma.reshape(1, -1, 1, 1) / n)
    return dy, dgamma, dbeta


def bn_train_fwd_stored(y, gamma, beta, storage):
    """bn_train_fwd for a STORED (already rounded) 16-bit tensor, the way the HIP kernels form the pre-activation: ONE f32 fused
    multiply-add of the stored y with f32 coefficients derived in double (torch_vae_amd/csrc/common.cuh bn_fused_channel:
    sc = gamma*invstd, sh = beta - mean*sc).  Where z lands within rounding of 0 this - not the exact value - decides LeakyReLU's
    slope, so the storage emulation takes the same route.  storage None: plain bn_train_fwd."""
    z, cache = bn_train_fwd(y, gamma, beta)
    if storage is not None:
        invstd, mean = cache[1].astype(np.float64), cache[2].astype(np.float64)
        sc64 = gamma.astype(np.float64) * invstd
        sc32 = sc64.astype(np.float32).astype(np.float64)
        sh32 = (beta.astype(np.float64) - mean * sc64).astype(np.float32).astype(np.float64)
        y32 = y.astype(np.float32).astype(np.float64)
        z = (y32 * sc32.reshape(1, -1, 1, 1) + sh32.reshape(1, -1, 1, 1)).astype(np.float32).astype(y.dtype)
    return z, cache


def bn_eval_fwd(y, gamma, beta, rm, rv):
    invstd = 1.0 / np.sqrt(rv + y.dtype.type(BN_EPS))
    return (y - rm.reshape(1, -1, 1, 1)) * (gamma * invstd).reshape(1, -1, 1, 1) + beta.reshape(1, -1, 1, 1)


def bn_eval_fwd_stored(y, gamma, beta, rm, rv, storage):
    """bn_eval_fwd the way the HIP kernels form it (torch_vae_amd/csrc/edge_kernels.cuh bn_eval_coef_kernel + the consumer's staging):
    ONE f32 fused multiply-add with f32 coefficients sc = gamma / sqrt(rv + eps), sh = beta - rm * sc derived in double.
    storage None: plain bn_eval_fwd."""
    if storage is None:
        return bn_eval_fwd(y, gamma, beta, rm, rv)
    sc64 = gamma.astype(np.float64) / np.sqrt(rv.astype(np.float64) + BN_EPS)
    sc32 = sc64.astype(np.float32).astype(np.float64)
    sh32 = (beta.astype(np.float64) - rm.astype(np.float64) * sc64).astype(np.float32).astype(np.float64)
    y32 = y.astype(np.float32).astype(np.float64)
    return (y32 * sc32.reshape(1, -1, 1, 1) + sh32.reshape(1, -1, 1, 1)).astype(np.float32).astype(y.dtype)


def lrelu(z):
    return np.where(z > 0, z, z * z.dtype.type(LEAKY_SLOPE))


def lrelu_bwd(z, da):
    return np.where(z > 0, da, da * z.dtype.type(LEAKY_SLOPE))


def sigmoid(v):
    return 1.0 / (1.0 + np.exp(-v))


# --------------------------------------------------------------------------
# storage emulation (16-bit kernel modes)
# --------------------------------------------------------------------------
F16_FLUSH_SUBNORMALS = False


def round_storage(a: np.ndarray, storage: str | None, scale: float = 1.0) -> np.ndarray:
    """Round to the grid of the HIP path's 16-bit storage type ("bf16" / "f16"; None: identity), round-to-nearest-even, keeping
    the array's dtype.  `scale` (a power of two): the f16 mode stores its gradients multiplied by it (torch_vae_amd/csrc/vae_ctx.h:
    gmul), so they are rounded on the scaled values."""
    if storage is None:
        return a
    v = a * a.dtype.type(scale) if scale != 1.0 else a
    if storage == "f16":
        with np.errstate(over="ignore"):
            r = v.astype(np.float16).astype(a.dtype)
        if F16_FLUSH_SUBNORMALS:          # (experiment switch, tools/diag/gpu_emu_gaps.py: results below 2^-14 become zero)
            r = np.where(np.abs(r) < 2.0 ** -14, r * 0, r)
    elif storage == "bf16":
        f = np.ascontiguousarray(v, dtype=np.float32)
        u = f.view(np.uint32)
        with np.errstate(over="ignore"):
            u = ((u + np.uint32(0x7FFF) + ((u >> np.uint32(16)) & np.uint32(1))) & np.uint32(0xFFFF0000)).astype(np.uint32)
        r = u.view(np.float32).astype(a.dtype)
    else:
        raise ValueError(storage)
    return r / a.dtype.type(scale) if scale != 1.0 else r


def f16_grad_scale(batch: int, img_size: int) -> float:
    """Gradient scale of the f16 kernel mode (vae_impl.cuh forward_impl: 2^(ilog2 B + 2 ilog2 H - 4), ilog2 = ceil(log2))."""
    il = lambda v: int(math.ceil(math.log2(v))) if v > 1 else 0   # noqa: E731
    return float(2 ** max(0, il(batch) + 2 * il(img_size) - 4))


# --------------------------------------------------------------------------
# model forward / loss / backward
# --------------------------------------------------------------------------
_ENC = [f"encoder.{i}" for i in range(4)]
_DEC = [f"decoder.{i}" for i in range(3)]


def forward(p: dict, x: np.ndarray, eps: np.ndarray, bn_state: dict | None = None,
            train: bool = True, update_running: bool = True, storage: str | None = None) -> dict:
    """models.py:185-188 (encode :107-145, reparameterize :177-183, decode :147-175).

    eps is the N(0,1) draw that torch.randn_like would supply (H5: passed
    explicitly for parity).  Returns a dict with every intermediate needed by
    backward plus the ModelOutput fields.

    storage ("bf16" / "f16"): emulate the rounding points of the HIP path's 16-bit kernel modes: every raw
    conv output y_l is STORED rounded (BatchNorm statistics are those of the stored tensor), the staged operand LeakyReLU(BN(y_l))
    is rounded when it enters a matrix product, the weights of the MFMA layers are rounded (packed images), decoder_input's
    output is stored rounded; accumulation, statistics, the latent block and the loss stay in the oracle's precision.
    """
    dt = x.dtype
    c = {"x": x, "eps": eps, "storage": storage}
    a = x
    rs = lambda v: round_storage(v, storage)   # noqa: E731

    def bn(name, y):
        g, b = p[name + ".weight"], p[name + ".bias"]
        if train:
            z, cache = bn_train_fwd_stored(y, g, b, storage)
            if bn_state is not None and update_running:
                n = y.shape[0] * y.shape[2] * y.shape[3]
                mean, var = cache[2], cache[3]
                bn_state[name + ".running_mean"] = (1 - BN_MOMENTUM) * bn_state[name + ".running_mean"] + BN_MOMENTUM * mean
                bn_state[name + ".running_var"] = (1 - BN_MOMENTUM) * bn_state[name + ".running_var"] + BN_MOMENTUM * var * n / max(n - 1, 1)
                bn_state[name + ".num_batches_tracked"] = bn_state[name + ".num_batches_tracked"] + 1
            return z, cache
        z = bn_eval_fwd(y, g, b, bn_state[name + ".running_mean"].astype(dt), bn_state[name + ".running_var"].astype(dt))
        return z, None

    for i, name in enumerate(_ENC):
        c[name + ".in"] = a                      # (already rounded where the HIP path stages it rounded)
        w = p[name + ".0.weight"] if i == 0 else rs(p[name + ".0.weight"])   # encoder.0 reads the f32 weights (conv1_fwd)
        y = rs(conv_fwd(a, w, p[name + ".0.bias"], 2))
        z, cache = bn(name + ".1", y)
        c[name + ".y"], c[name + ".z"], c[name + ".bn"] = y, z, cache
        c[name + ".act"] = lrelu(z)              # unrounded (the fc weight gradient / pre_latents use it in f32)
        a = rs(c[name + ".act"])
    B = x.shape[0]
    c["enc_shape"] = a.shape
    pre = a.reshape(B, -1)                       # staged (rounded) operand of the fc products
    mu = pre @ rs(p["fc_mu.weight"]).T + p["fc_mu.bias"]
    lv = pre @ rs(p["fc_var.weight"]).T + p["fc_var.bias"]
    std = np.exp(dt.type(0.5) * lv)
    zlat = eps * std + mu
    d0 = rs(zlat @ p["decoder_input.weight"].T + p["decoder_input.bias"])
    s = int(round(math.sqrt(d0.shape[1] // 256)))
    a = d0.reshape(-1, 256, s, s)  # models.py:166 (generalised: s = H/16)
    c.update(pre=c[_ENC[-1] + ".act"].reshape(B, -1), mu=mu, lv=lv, std=std, zlat=zlat)
    for name in _DEC:
        c[name + ".in"] = a
        y = rs(convT_fwd(a, rs(p[name + ".0.weight"]), p[name + ".0.bias"]))
        z, cache = bn(name + ".1", y)
        c[name + ".y"], c[name + ".z"], c[name + ".bn"] = y, z, cache
        a = rs(lrelu(z))
    c["final_layer.in"] = a
    y = rs(convT_fwd(a, rs(p["final_layer.0.weight"]), p["final_layer.0.bias"]))
    z, cache = bn("final_layer.1", y)
    c["final_layer.y"], c["final_layer.z"], c["final_layer.bn"] = y, z, cache
    a = rs(lrelu(z))
    c["final_conv.in"] = a
    logits = conv_fwd(a, rs(p["final_layer.3.weight"]), p["final_layer.3.bias"], 1)
    c["logits"] = logits
    c["output"] = sigmoid(logits)
    return c


def loss(c: dict, kld_weight: float = 1.0) -> dict:
    """models.py:190-225.  BCE follows ATen: log terms clamped at -100."""
    xh, t = c["output"], c["x"]
    dt = xh.dtype
    with np.errstate(divide="ignore"):
        l1 = np.maximum(np.log(xh), dt.type(-100.0))
        l0 = np.maximum(np.log1p(-xh) if dt == np.float64 else np.log(dt.type(1.0) - xh), dt.type(-100.0))
    recon = np.mean(-(t * l1 + (1 - t) * l0), dtype=np.float64 if dt == np.float64 else dt)
    mu, lv = c["mu"], c["lv"]
    kld = dt.type(-0.5) * np.mean(np.sum(1 + lv - mu ** 2 - np.exp(lv), axis=-1))
    total = recon + dt.type(kld_weight) * kld
    return {"loss": total, "reconstruction_loss": recon, "kld_loss": -kld}


def backward(p: dict, c: dict, kld_weight: float = 1.0) -> dict:
    """Analytic gradient of loss()['loss'] w.r.t. every parameter (train.py:650)."""
    dt = c["output"].dtype
    g = {}
    xh, t = c["output"], c["x"]
    N = xh.size
    # storage emulation (see forward): stored gradients dz_l and the staged BatchNorm-backward gradients are rounded (on values
    # multiplied by the f16 mode's gradient scale), gradient products use the rounded weights; sums stay in the oracle's precision
    storage = c.get("storage")
    gs = f16_grad_scale(xh.shape[0], xh.shape[2]) if storage == "f16" else 1.0
    rs = lambda v: round_storage(v, storage)            # noqa: E731
    rg = lambda v: round_storage(v, storage, gs)        # noqa: E731
    # F.binary_cross_entropy backward (ATen): (x-t)/max(x(1-x),1e-12) * grad
    d_xh = (xh - t) / np.maximum(xh * (1 - xh), dt.type(1e-12)) / dt.type(N)
    dlogit = d_xh * xh * (1 - xh)
    a = c["final_conv.in"]
    dl_op = rg(dlogit)                                  # the MFMA operand of the output conv's gradient products
    dw, _ = conv_wgrad(a, dl_op, 1)
    g["final_layer.3.weight"], g["final_layer.3.bias"] = dw, dlogit.sum(axis=(0, 2, 3))
    da = conv_dgrad(dl_op, rs(p["final_layer.3.weight"]), 1, a.shape[2:])

    def block_bwd(name, bn_name, da, transposed):
        dz = rg(lrelu_bwd(c[name + ".z"], da))
        g[name + ".dz"] = dz
        dy, dgam, dbet = bn_train_bwd(dz, p[bn_name + ".weight"], c[name + ".bn"])
        g[bn_name + ".weight"], g[bn_name + ".bias"] = dgam, dbet
        first = name == _ENC[0]                         # encoder.0: f32 weights and an unrounded gradient operand (conv1_wgrad)
        dyr = dy if first else rg(dy)
        xin = c[name + ".in"]
        w = p[name + ".0.weight"] if first else rs(p[name + ".0.weight"])
        if transposed:
            dx, dw, db = convT_bwd(xin, w, dyr)
        else:
            dw, db = conv_wgrad(xin, dyr, 2)
            dx = conv_dgrad(dyr, w, 2, xin.shape[2:])
        g[name + ".0.weight"], g[name + ".0.bias"] = dw, db
        return dx

    da = block_bwd("final_layer", "final_layer.1", da, True)
    for name in reversed(_DEC):
        da = block_bwd(name, name + ".1", da, True)
    B = da.shape[0]
    dd0 = rg(da).reshape(B, -1)                         # stored by decoder.0's input-gradient kernel
    g["__dd0"] = dd0
    g["decoder_input.weight"] = dd0.T @ c["zlat"]
    g["decoder_input.bias"] = dd0.sum(axis=0)
    dzlat = dd0 @ rs(p["decoder_input.weight"])
    mu, lv = c["mu"], c["lv"]
    kw = dt.type(kld_weight)
    dmu = dzlat + kw * mu / B
    dlv = dzlat * c["eps"] * c["std"] * dt.type(0.5) + kw * dt.type(0.5) * (np.exp(lv) - 1) / B
    g["fc_mu.weight"] = dmu.T @ c["pre"]
    g["fc_mu.bias"] = dmu.sum(axis=0)
    g["fc_var.weight"] = dlv.T @ c["pre"]
    g["fc_var.bias"] = dlv.sum(axis=0)
    dpre = dmu @ rs(p["fc_mu.weight"]) + dlv @ rs(p["fc_var.weight"])
    da = dpre.reshape(c["enc_shape"])
    for name in reversed(_ENC):
        da = block_bwd(name, name + ".1", da, False)
    g["__dmu"], g["__dlv"], g["__dlogit"] = dmu, dlv, dlogit
    return g


# --------------------------------------------------------------------------
# optimiser + scheduler (train.py:201-238)
# --------------------------------------------------------------------------
def optimised_names(shapes) -> tuple[list, list]:
    """Only model.encoder and model.decoder are in the optimiser (train.py:210-225)."""
    enc = [n for n in shapes if n.startswith("encoder.")]
    dec = [n for n in shapes if n.startswith("decoder.")]
    return enc, dec


def scaled_lr(lr_relative: float, batch_size_per_gpu: int, world_size: int = 1) -> float:
    """train.py:165-166, 201: lr = lr_relative * (per_gpu * world) / 128."""
    return lr_relative * batch_size_per_gpu * world_size / 128.0


@dataclass
class OneCycle:
    """torch.optim.lr_scheduler.OneCycleLR defaults as used at train.py:233-238:
    pct_start 0.3, cos, div_factor 25, final_div 1e4, base/max momentum 0.85/0.95,
    two phases.  value(step) is what the optimiser uses for its (step+1)-th update."""
    max_lr: float
    total_steps: int
    pct_start: float = 0.3
    div_factor: float = 25.0
    final_div_factor: float = 1e4
    base_momentum: float = 0.85
    max_momentum: float = 0.95

    @staticmethod
    def _cos(start, end, pct):
        return end + (start - end) / 2.0 * (math.cos(math.pi * pct) + 1)

    def value(self, step_num: int) -> tuple[float, float]:
        if step_num > self.total_steps:
            raise ValueError("Tried to step more than total_steps")  # torch raises too
        init_lr = self.max_lr / self.div_factor
        min_lr = init_lr / self.final_div_factor
        e1 = float(self.pct_start * self.total_steps) - 1
        e2 = self.total_steps - 1
        if step_num <= e1:
            pct = step_num / e1
            return self._cos(init_lr, self.max_lr, pct), self._cos(self.max_momentum, self.base_momentum, pct)
        pct = (step_num - e1) / (e2 - e1)
        return self._cos(self.max_lr, min_lr, pct), self._cos(self.base_momentum, self.max_momentum, pct)


@dataclass
class AdamWState:
    m: dict = field(default_factory=dict)
    v: dict = field(default_factory=dict)
    step: int = 0


def adamw_update(p, g, st: AdamWState, names, lr, beta1, beta2=0.999, eps=1e-8, weight_decay=0.0):
    """torch.optim.AdamW single step on `names` (decoupled decay, train.py:228)."""
    t = st.step
    bc1 = 1 - beta1 ** t
    bc2 = 1 - beta2 ** t
    for n in names:
        dt = p[n].dtype
        if n not in st.m:
            st.m[n] = np.zeros_like(p[n])
            st.v[n] = np.zeros_like(p[n])
        p[n] = p[n] * dt.type(1 - lr * weight_decay)
        st.m[n] = st.m[n] * dt.type(beta1) + dt.type(1 - beta1) * g[n]
        st.v[n] = st.v[n] * dt.type(beta2) + dt.type(1 - beta2) * g[n] * g[n]
        denom = np.sqrt(st.v[n]) / dt.type(math.sqrt(bc2)) + dt.type(eps)
        p[n] = p[n] - dt.type(lr / bc1) * (st.m[n] / denom)


@dataclass
class Trainer:
    """train_one_epoch's loop body (train.py:620-664) on oracle state."""
    p: dict
    bn_state: dict
    sched_enc: OneCycle
    sched_dec: OneCycle
    kld_weight: float = 1.0
    weight_decay: float = 0.0
    opt: AdamWState = field(default_factory=AdamWState)
    sched_step: int = 0

    def step(self, x, eps, grads_override=None):
        c = forward(self.p, x, eps, self.bn_state, train=True)
        lo = loss(c, self.kld_weight)
        g = backward(self.p, c, self.kld_weight)
        if grads_override is not None:
            g = grads_override(g)
        enc, dec = optimised_names(self.p)
        lr_e, b1_e = self.sched_enc.value(self.sched_step)
        lr_d, b1_d = self.sched_dec.value(self.sched_step)
        self.opt.step += 1
        adamw_update(self.p, g, self.opt, enc, lr_e, b1_e, weight_decay=self.weight_decay)
        adamw_update(self.p, g, self.opt, dec, lr_d, b1_d, weight_decay=self.weight_decay)
        self.sched_step += 1
        return lo, c, g


def make_trainer(latent_dim, img_size, batch, total_steps, seed=0, generalised=False,
                 kld_weight=1.0, lr_relative=0.01, world_size=1, dtype=np.float64) -> Trainer:
    p = init_params(latent_dim, img_size, seed, generalised, dtype)
    lr = scaled_lr(lr_relative, batch, world_size)
    return Trainer(p, init_bn_state(dtype), OneCycle(lr, total_steps), OneCycle(lr, total_steps), kld_weight)
