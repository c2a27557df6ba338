"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle and the golden
fixtures generated from the reference.  Tolerances:
  f32 kernel mode : ELBO scalars and forward tensors <= 1e-4 relative (BASELINE.json north_star).
                    Gradients typically agree to ~1e-6 rel-L2, but LeakyReLU makes them discontinuous:
                    one pre-activation within rounding distance of 0 (binary inputs make exact ties
                    common) flips a derivative and moves a whole gradient by ~1e-3, so gradient
                    checks allow 5e-3 ("kink ties").
  bf16 kernel mode: ELBO scalars <= 1e-2 relative (measured <= 5.3e-3; the reference's own bf16 autocast gap is 4.4e-3,
                    SURVEY.md H4); gradients: see GRAD_TOL / GRAD_TOL_FULL / GRAD_TOL_FIXTURE below.
  f16 kernel mode : ELBO scalars <= 2e-3 (measured <= 6.3e-4), gradients likewise (10 mantissa bits instead of 7).
The 16-bit bounds are what the kernels measure on MI355X plus headroom (every run appends its measured gaps to
gpurun_out/parity_report.jsonl); they bound direction AND size of every gradient tensor.
"""
import json
import os

import numpy as np
import pytest
import torch

from oracle import vae_oracle as vo
from tests.util import CASES, PRE_BN_BIAS, flat_grad_dict, load_params, make_model, perturbed_params, rel_l2

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ELBO_TOL = {"f32": 1e-4, "bf16": 1e-2, "f16": 2e-3}
# per-tensor gradient rel-L2 vs the fp64 oracle.  16-bit storage of the pre-activations flips LeakyReLU's slope for the
# elements that sit within one rounding step of zero (the gradient is discontinuous there, factor 100): the error of a
# gradient tensor is ~sqrt(fraction of flipped elements), independent of how many pixels it sums over.  Measured on MI355X
# over the adversarial small cases below (non-trivial BatchNorm affine, batches of 1-33): bf16 0.11-0.21, f16 0.03-0.10;
# at the BASELINE workload sizes the whole flat gradient agrees with the f32 mode to 6e-3 (bf16) / 2.4e-3 (f16) - gated in
# test_full_size_baseline_configs - and the reference-initialised fixtures to 5e-2 / 3e-2 per tensor norm.
GRAD_TOL = {"f32": 5e-3, "bf16": 0.28, "f16": 0.14}
GRAD_TOL_FULL = {"bf16": 2e-2, "f16": 8e-3}              # flat gradient vs the f32 mode at full size
GRAD_TOL_FIXTURE = {"bf16": 8e-2, "f16": 4e-2}           # per-tensor gradient NORM vs the reference fixture
FWD_TOL = {"f32": 1e-4, "bf16": 2e-2, "f16": 3e-3}        # xhat rel-L2
# against the oracle run with the kernels' storage rounding emulated (same rounding points): per-tensor gradient rel-L2 / ELBO
# Measured on MI355X over the 20 adversarial small cases x 31 tensors (gpurun_out/parity_report.jsonl, test "every_tensor_vs_emulated_storage"):
# bf16: the worst tensor of a case sits at 0.004-0.06 (median 0.009; three cases at 0.06-0.093), where the same steps measure 0.11-0.21 against the
# exact oracle; whole cases agree to 0.000-0.005 on every tensor (tools/diag/gpu_emu_gaps.py).  f16: 0.02-0.07 (median 0.03) against 0.03-0.10.
# What is left was traced (tools/diag/gpu_emu_dz.py): f32-in-MFMA-order vs f64 accumulation makes a growing share of the stored values differ
# by one ulp (26 % of y7 at final_layer), and the ~0.04 % of elements whose pre-activation lies within one storage ulp of zero then take the
# other LeakyReLU slope: they carry 99 % of dz7's 2.8 % mismatch, which travels down the decoder.  Not reproducible by any CPU emulation.
EMU_GRAD_TOL = {"bf16": 0.12, "f16": 0.09}
EMU_ELBO_TOL = {"bf16": 2e-3, "f16": 5e-4}


def report(**kw):
    """Measured gaps of the 16-bit modes, kept under gpurun_out/ (scratch) so the bounds above can be audited."""
    try:
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        with open(os.path.join(ROOT, "gpurun_out", "parity_report.jsonl"), "a") as f:
            f.write(json.dumps(kw) + "\n")
    except OSError:
        pass



def case_inputs(name, step):
    H, L, B, steps, total, kw, gen, seed = CASES[name]
    x = vo.synth_pianoroll(B, H, seed * 1000 + step)
    eps = vo.counter_normal(B * L, seed * 1000 + step, 5).reshape(B, L)
    return x, eps


def test_tr16_selftest():
    from torch_vae_amd import _lib
    rc = _lib.lib().vae_selftest_tr16(torch.cuda.current_stream().cuda_stream)
    assert rc == 0, _lib.lib().vae_last_error().decode()


@pytest.mark.parametrize("name", list(CASES))
def test_f32_step0_matches_reference_fixture(name):
    """forward + ELBO + gradients of step 0 against the REFERENCE's own outputs (golden fixtures)."""
    H, L, B, steps, total, kw, gen, seed = CASES[name]
    gold = np.load(os.path.join(GOLD, f"{name}_f64.npz"))
    model = make_model(H, L, gen, "f32", vo.init_params(L, H, seed, gen), kld_weight=kw)
    x, eps = case_inputs(name, 0)
    out3, xhat = model.fused_forward_backward(torch.from_numpy(x).cuda(), eps=torch.from_numpy(eps).float().cuda())
    got = np.array(out3.tolist())
    np.testing.assert_allclose(got, gold["losses"][0], rtol=1e-4)   # the 1e-4 ELBO gate
    assert rel_l2(model._last["mu"].cpu().numpy(), gold["mu"]) < 1e-4
    assert rel_l2(model._last["lv"].cpu().numpy(), gold["log_var"]) < 1e-4
    assert rel_l2(model._last["z"].cpu().numpy(), gold["latents"]) < 1e-4
    xh = xhat.double().cpu().numpy()
    assert abs(xh.sum() - float(gold["output_sum"])) < 1e-5 * abs(float(gold["output_sum"]))
    np.testing.assert_allclose(xh.reshape(-1)[gold["output_idx"]], gold["output_samples"], rtol=2e-4, atol=1e-6)
    g = flat_grad_dict(model)
    for n, v in g.items():
        ref = float(gold["gradnorm/" + n])
        if n in PRE_BN_BIAS:
            assert np.abs(v).max() < 1e-6
            continue
        assert abs(np.sqrt((v.astype(np.float64) ** 2).sum()) - ref) < 5e-3 * ref + 1e-9, n
        gi = (vo.counter_uniform(16, seed, 123) * v.size).astype(np.int64)
        np.testing.assert_allclose(v[gi], gold["gradsamp/" + n], rtol=1e-2, atol=1e-2 * ref / np.sqrt(v.size) + 1e-9, err_msg=n)  # atol = 1% of the tensor RMS


@pytest.mark.parametrize("cfg", [(32, 16, 5, False), (32, 16, 33, False), (64, 32, 3, True), (128, 16, 2, True),
                                 (32, 16, 1, False), (64, 8, 1, True), (32, 128, 7, True), (256, 16, 1, True), (32, 4, 130, False),
                                 # latent sizes whose padded widths are not a power-of-two number of 32-column blocks
                                 # (npad 96 / 160 / 192: the dense kernel's N tiling) and sizes that are not a multiple of 4
                                 (32, 40, 3, False), (32, 48, 5, False), (32, 80, 3, False), (32, 96, 4, False), (64, 160, 2, True),
                                 (32, 10, 6, False), (64, 10, 2, True), (32, 1, 3, False), (32, 3, 9, False),
                                 (32, 16, 256, False), (64, 16, 48, True)])
@pytest.mark.parametrize("dtype", ["f32", "bf16", "f16"])
def test_every_tensor_against_oracle(cfg, dtype):
    """All gradients (full tensors) against the fp64 oracle, with non-trivial BN affine/bias values,
    ragged batch sizes (not a multiple of any tile), batch 1, the smallest / largest latent sizes and image sizes,
    the reference's default latent size 10 (train.py:875-877)."""
    H, L, B, gen = cfg
    p = perturbed_params(L, H, 17, gen)
    model = make_model(H, L, gen, dtype, p)
    x = vo.synth_pianoroll(B, H, 3)
    eps = vo.counter_normal(B * L, 3, 5).reshape(B, L)
    out3, xhat = model.fused_forward_backward(torch.from_numpy(x).cuda(), eps=torch.from_numpy(eps).float().cuda())
    c = vo.forward(p, x.astype(np.float64), eps, None, train=True)
    lo = vo.loss(c)
    g = vo.backward(p, c)
    want = np.array([float(lo["loss"]), float(lo["reconstruction_loss"]), float(lo["kld_loss"])])
    got3 = np.array(out3.tolist())
    got = flat_grad_dict(model)
    gaps = {n: rel_l2(v, g[n].reshape(-1)) for n, v in got.items() if n not in PRE_BN_BIAS}
    worst = max(gaps, key=gaps.get)
    report(test="every_tensor", cfg=list(cfg), dtype=dtype, elbo_rel=float(np.abs(got3 / want - 1).max()),
           xhat_rel_l2=rel_l2(xhat.cpu().numpy(), c["output"]), grad_rel_l2_max=gaps[worst], worst=worst)
    np.testing.assert_allclose(got3, want, rtol=ELBO_TOL[dtype])
    assert rel_l2(xhat.cpu().numpy(), c["output"]) < FWD_TOL[dtype]
    assert rel_l2(model._last["mu"].cpu().numpy(), c["mu"]) < 1.5 * FWD_TOL[dtype]
    for n, gap in gaps.items():
        assert gap < GRAD_TOL[dtype], (n, gap)
    if dtype != "f32":
        # The same step against the oracle with the kernels' STORAGE emulated (y_l, dz_l, the staged operands and the packed weights
        # rounded where the kernels round them: oracle/vae_oracle.py forward/backward(storage=...)).  This separates the two
        # error sources the wide gate above lumps together: what remains here is the kernels' own arithmetic (accumulation order,
        # f32 instead of f64 statistics), so the gate is tight.
        ce = vo.forward(p, x.astype(np.float64), eps, None, train=True, storage=dtype)
        le = vo.loss(ce)
        ge = vo.backward(p, ce)
        wante = np.array([float(le["loss"]), float(le["reconstruction_loss"]), float(le["kld_loss"])])
        gaps_e = {n: rel_l2(v, ge[n].reshape(-1)) for n, v in got.items() if n not in PRE_BN_BIAS}
        worst_e = max(gaps_e, key=gaps_e.get)
        report(test="every_tensor_vs_emulated_storage", cfg=list(cfg), dtype=dtype, elbo_rel=float(np.abs(got3 / wante - 1).max()),
               xhat_rel_l2=rel_l2(xhat.cpu().numpy(), ce["output"]), grad_rel_l2_max=gaps_e[worst_e], worst=worst_e)
        np.testing.assert_allclose(got3, wante, rtol=EMU_ELBO_TOL[dtype])
        for n, gap in gaps_e.items():
            assert gap < EMU_GRAD_TOL[dtype], (n, gap, "vs the storage-emulating oracle")


@pytest.mark.parametrize("cfg", [(128, 16, 3, "bf16"), (128, 16, 2, "f16"), (256, 16, 1, "bf16"), (128, 64, 5, "f16")])
def test_deep_layer_kernels_against_oracle(cfg):
    """The workgroup-specialised deep-layer kernels (conv_deep.cuh: dn3 for the stride-2 products, up3 for the transposed ones;
    up3 is off by default) switched on together, on the shapes they engage (128x128 and 256x256 images: 8x8 / 16x16 / 32x32
    low-res maps, ragged batches): every gradient tensor against the storage-emulating oracle, and against the pipelined kernels."""
    from torch_vae_amd import _lib
    H, L, B, dtype = cfg
    p = perturbed_params(L, H, 23, True)
    x = vo.synth_pianoroll(B, H, 13)
    eps = vo.counter_normal(B * L, 13, 5).reshape(B, L)
    ce = vo.forward(p, x.astype(np.float64), eps, None, train=True, storage=dtype)
    le = vo.loss(ce)
    ge = vo.backward(p, ce)
    want = np.array([float(le["loss"]), float(le["reconstruction_loss"]), float(le["kld_loss"])])
    res = {}
    for deep in (3, 0):
        model = make_model(H, L, True, dtype, p)
        model._context(B)
        assert _lib.lib().vae_set_option(model._ctx.handle, b"use_deep", deep) == 0
        out3, _ = model.fused_forward_backward(torch.from_numpy(x).cuda(), eps=torch.from_numpy(eps).float().cuda())
        got = flat_grad_dict(model)
        gaps = {n: rel_l2(v, ge[n].reshape(-1)) for n, v in got.items() if n not in PRE_BN_BIAS}
        worst = max(gaps, key=gaps.get)
        report(test="deep_kernels_vs_emulated_storage", cfg=list(cfg), use_deep=deep, elbo_rel=float(np.abs(np.array(out3.tolist()) / want - 1).max()),
               grad_rel_l2_max=gaps[worst], worst=worst)
        np.testing.assert_allclose(np.array(out3.tolist()), want, rtol=EMU_ELBO_TOL[dtype])
        for n, gap in gaps.items():
            assert gap < EMU_GRAD_TOL[dtype], (deep, n, gap)
        res[deep] = (np.array(out3.tolist()), got)
    np.testing.assert_allclose(res[3][0], res[0][0], rtol=EMU_ELBO_TOL[dtype])


def test_tr16_and_scalar_wgrad_agree():
    from torch_vae_amd import _lib
    H, L, B, gen = 64, 16, 6, True
    p = perturbed_params(L, H, 5, gen)
    x = torch.from_numpy(vo.synth_pianoroll(B, H, 9)).cuda()
    eps = torch.from_numpy(vo.counter_normal(B * L, 9, 5).reshape(B, L)).float().cuda()
    grads = []
    for tr in (0, 1):
        model = make_model(H, L, gen, "bf16", p)
        model._context(B)
        assert _lib.lib().vae_set_option(model._ctx.handle, b"use_tr16", tr) == 0
        model.fused_forward_backward(x, eps=eps)
        grads.append(model.flat_grads().clone())
    # same bf16 operands, same k order inside each MFMA: only the split of K over waves can differ
    assert rel_l2(grads[1].cpu().numpy(), grads[0].cpu().numpy()) < 1e-5


def test_mfma_and_valu_output_conv_backward_agree():
    """bf16 mode: the MFMA output-conv backward against the f32-VALU formulation of the same math."""
    from torch_vae_amd import _lib
    H, L, B, gen = 64, 16, 5, True
    p = perturbed_params(L, H, 8, gen)
    x = torch.from_numpy(vo.synth_pianoroll(B, H, 12)).cuda()
    eps = torch.from_numpy(vo.counter_normal(B * L, 12, 5).reshape(B, L)).float().cuda()
    grads = []
    for use in (0, 1):
        model = make_model(H, L, gen, "bf16", p)
        model._context(B)
        assert _lib.lib().vae_set_option(model._ctx.handle, b"use_mfma_convout", use) == 0
        model.fused_forward_backward(x, eps=eps)
        grads.append(flat_grad_dict(model))
    for n in grads[0]:
        if n in PRE_BN_BIAS:
            continue
        a, b = grads[1][n].astype(np.float64), grads[0][n].astype(np.float64)
        cos = float(np.dot(a, b) / (np.linalg.norm(a) * np.linalg.norm(b) + 1e-30))
        assert cos > 0.995, (n, cos)
    assert rel_l2(grads[1]["final_layer.3.weight"], grads[0]["final_layer.3.weight"]) < 2e-2
    assert rel_l2(grads[1]["final_layer.3.bias"], grads[0]["final_layer.3.bias"]) < 2e-3
    assert rel_l2(grads[1]["final_layer.1.weight"], grads[0]["final_layer.1.weight"]) < 1e-2


@pytest.mark.parametrize("name", ["R_b32_k1", "G_h64_l16_b4"])
def test_f32_training_trajectory(name):
    """Loss curve, parameters and BN running statistics over several fused steps (forward, ELBO,
    backward, AdamW, OneCycle) against the reference's trajectory."""
    from torch_vae_amd.optim import FusedAdamW
    H, L, B, steps, total, kw, gen, seed = CASES[name]
    gold = np.load(os.path.join(GOLD, f"{name}_f64.npz"))
    model = make_model(H, L, gen, "f32", vo.init_params(L, H, seed, gen), kld_weight=kw)
    lr = vo.scaled_lr(0.01, B)
    opt = FusedAdamW([{"params": model.encoder.parameters(), "lr": lr}, {"params": model.decoder.parameters(), "lr": lr}],
                     lr=lr, weight_decay=0.0)
    sched = torch.optim.lr_scheduler.OneCycleLR(opt, [lr, lr], epochs=1, steps_per_epoch=total)
    n = min(steps, 8)
    losses = []
    for s in range(n):
        x, eps = case_inputs(name, s)
        out3, _ = model.fused_forward_backward(torch.from_numpy(x).cuda(), eps=torch.from_numpy(eps).float().cuda())
        opt.step()
        sched.step()
        losses.append(out3.tolist())
    np.testing.assert_allclose(np.array(losses), gold["losses"][:n], rtol=3e-4)
    if n == steps:
        sd = model.state_dict()
        for k in gold.files:
            if k.startswith("buf/") and not k.endswith("num_batches_tracked"):
                np.testing.assert_allclose(sd[k[4:]].cpu().numpy(), gold[k], rtol=1e-3, atol=1e-5, err_msg=k)
            if k.startswith("buf/") and k.endswith("num_batches_tracked"):
                assert int(sd[k[4:]]) == int(gold[k])
            if k.startswith("param_l2/") and k[9:] not in PRE_BN_BIAS:
                v = sd[k[9:]].double().cpu().numpy()
                np.testing.assert_allclose(np.sqrt((v ** 2).sum()), gold[k], rtol=1e-3, err_msg=k)


def test_autograd_path_matches_fused_path():
    """model.forward -> zero_grad -> model.loss -> backward (the reference's call sequence,
    train.py:634-650) gives the same numbers as the fused chain."""
    H, L, B, gen = 32, 16, 16, False
    p = perturbed_params(L, H, 2, gen)
    x = torch.from_numpy(vo.synth_pianoroll(B, H, 4)).cuda()
    eps = torch.from_numpy(vo.counter_normal(B * L, 4, 5).reshape(B, L)).float().cuda()
    m1 = make_model(H, L, gen, "f32", p, kld_weight=4.0)
    out3, _ = m1.fused_forward_backward(x, eps=eps)
    m2 = make_model(H, L, gen, "f32", p, kld_weight=4.0)
    m2.set_next_eps(eps)
    out = m2.forward(x)
    assert set(out) == {"output", "input", "encoded", "latents"} and set(out["encoded"]) == {"mu", "log_var", "pre_latents"}
    assert out["encoded"]["pre_latents"].shape == (B, 1024)
    lo = m2.loss(out)
    assert set(lo) == {"loss", "reconstruction_loss", "kld_loss"}
    assert lo["loss"].requires_grad and not lo["reconstruction_loss"].requires_grad
    lo["loss"].backward()
    np.testing.assert_allclose([lo["loss"].item(), lo["reconstruction_loss"].item(), lo["kld_loss"].item()], out3.tolist(), rtol=1e-6)
    for (n, a), (_, b) in zip(m1.named_parameters(), m2.named_parameters()):
        assert b.grad is not None, n
        np.testing.assert_allclose(b.grad.cpu().numpy(), a.grad.cpu().numpy(), rtol=1e-5, atol=1e-8, err_msg=n)
    # second backward without zero_grad accumulates, as autograd does for the reference
    g1 = {n: q.grad.clone() for n, q in m2.named_parameters()}
    m2.set_next_eps(eps)
    out = m2.forward(x)
    (2.0 * m2.loss(out)["loss"]).backward()
    for n, q in m2.named_parameters():
        if n in PRE_BN_BIAS:
            continue
        # BN running stats do not enter train-mode outputs, so the second gradient is 2x the first
        np.testing.assert_allclose(q.grad.cpu().numpy(), 3.0 * g1[n].cpu().numpy(), rtol=2e-4, atol=1e-7, err_msg=n)


def test_generic_loss_and_extra_gradients():
    """loss() on foreign tensors uses the generic ELBO kernel; extra terms on mu flow through backward."""
    H, L, B, gen = 32, 16, 8, False
    p = perturbed_params(L, H, 3, gen)
    x = torch.from_numpy(vo.synth_pianoroll(B, H, 6)).cuda()
    eps = torch.from_numpy(vo.counter_normal(B * L, 6, 5).reshape(B, L)).float().cuda()
    m = make_model(H, L, gen, "f32", p)
    m.set_next_eps(eps)
    out = m.forward(x)
    foreign = {"output": out["output"] * 1.0, "input": x, "encoded": {"mu": out["encoded"]["mu"] * 1.0, "log_var": out["encoded"]["log_var"] * 1.0}}
    lg = m.loss(foreign)
    lf = m.loss(out)
    np.testing.assert_allclose(lg["loss"].item(), lf["loss"].item(), rtol=1e-6)
    np.testing.assert_allclose(lg["kld_loss"].item(), lf["kld_loss"].item(), rtol=1e-6)
    lg["loss"].backward()
    g_generic = m.flat_grads().clone()
    m2 = make_model(H, L, gen, "f32", p)
    m2.fused_forward_backward(x, eps=eps)
    assert rel_l2(g_generic.cpu().numpy(), m2.flat_grads().cpu().numpy()) < 1e-5


def test_eval_mode_uses_running_stats_and_edge_inputs():
    H, L, B, gen = 32, 16, 8, False
    p = perturbed_params(L, H, 4, gen)
    m = make_model(H, L, gen, "f32", p)
    st = vo.init_bn_state()
    rng = np.random.default_rng(0)
    sd = m.state_dict()
    for k in st:
        if k.endswith("running_mean"):
            st[k] = 0.1 * rng.standard_normal(st[k].shape)
        if k.endswith("running_var"):
            st[k] = 0.5 + rng.random(st[k].shape)
        if not k.endswith("num_batches_tracked"):
            sd[k] = torch.from_numpy(st[k]).float()
    m.load_state_dict(sd)
    m.eval()
    for x in (np.zeros((B, 1, H, H), np.float32), np.ones((B, 1, H, H), np.float32), vo.synth_pianoroll(B, H, 1)):
        eps = vo.counter_normal(B * L, 8, 5).reshape(B, L)
        m.set_next_eps(torch.from_numpy(eps).float().cuda())
        with torch.no_grad():
            out = m(torch.from_numpy(x).cuda())
            lo = m.loss(out)
        c = vo.forward(p, x.astype(np.float64), eps, st, train=False)
        want = vo.loss(c)
        assert rel_l2(out["output"].cpu().numpy(), c["output"]) < 1e-4
        np.testing.assert_allclose(lo["loss"].item(), float(want["loss"]), rtol=1e-4)
    assert int(m.state_dict()["encoder.0.1.num_batches_tracked"]) == 0  # eval never updates


def test_device_generators_match_oracle():
    from torch_vae_amd import _lib
    x = torch.empty(6, 1, 64, 64, device="cuda")
    _lib.check(_lib.lib().vae_synth_pianoroll(x.data_ptr(), 6, 64, 42, torch.cuda.current_stream().cuda_stream), "synth")
    np.testing.assert_array_equal(x.cpu().numpy(), vo.synth_pianoroll(6, 64, 42))
    m = make_model(32, 16, False, "f32", vo.init_params(16, 32, 1, False))
    m.eps_seed = 100
    xx = torch.from_numpy(vo.synth_pianoroll(4, 32, 1)).cuda()
    m._run_forward(xx, None, train=True)
    got = torch.empty(4, 16, device="cuda")
    _lib.check(_lib.lib().vae_last_eps(m._ctx.handle, got.data_ptr(), torch.cuda.current_stream().cuda_stream), "eps")
    np.testing.assert_allclose(got.cpu().numpy(), vo.counter_normal(64, 101, 5).reshape(4, 16), rtol=1e-6, atol=1e-7)


def test_errors_are_loud():
    m = make_model(32, 16, False, "f32")
    with pytest.raises(RuntimeError):
        m(torch.zeros(2, 1, 64, 64, device="cuda"))      # the reference raises a shape RuntimeError too
    from torch_vae_amd.models import VanillaVAE
    cpu_model = VanillaVAE(1, 16, 32)
    with pytest.raises(RuntimeError):
        cpu_model(torch.zeros(2, 1, 32, 32))             # no CPU fallback
    with pytest.raises(NotImplementedError):
        VanillaVAE(3, 16, 32)


def test_train_one_epoch_matches_reference_loop():
    """torch_vae_amd.train.train_one_epoch over a list 'dataloader' reproduces the reference's
    train_one_epoch trajectory (fixture R_b4_k1: steps 1..2 came from the reference's own loop)."""
    from argparse import Namespace
    from torch_vae_amd.train import build_optimizer, train_one_epoch
    name = "R_b4_k1"
    H, L, B, steps, total, kw, gen, seed = CASES[name]
    gold = np.load(os.path.join(GOLD, f"{name}_f64.npz"))
    model = make_model(H, L, gen, "f32", vo.init_params(L, H, seed, gen), kld_weight=kw)
    cfg = Namespace(batch_size_per_gpu=B, world_size=1, lr_relative=0.01, weight_decay=0.0, optimizer="AdamW", scheduler="OneCycle",
                    epochs=1, log_wandb=False, print_interval=1000, log_interval=1000, freeze_encoder=False, global_rank=0)
    opt, sched = build_optimizer(cfg, model, steps_per_epoch=total)
    epss = [torch.from_numpy(case_inputs(name, s)[1]).float().cuda() for s in range(steps)]
    loader = [(torch.from_numpy(case_inputs(name, s)[0]), torch.zeros(B, dtype=torch.long)) for s in range(steps)]
    it = iter(epss)
    # (the loop takes its step through the one-call path, VanillaVAE.fused_train_step: the reference's noise is injected there)
    orig = model.fused_train_step
    model.fused_train_step = lambda o, x, **k: orig(o, x, **{**k, "eps": next(it)})
    res, total_step, n_seen = train_one_epoch(cfg, model, opt, sched, model.loss, loader, device="cuda", epoch=2)
    assert total_step == steps and n_seen == steps * B
    np.testing.assert_allclose(res["loss"], gold["losses"][:, 0].mean(), rtol=2e-4)
    # the same loop through the five-call path (VAE_ONE_CALL_STEP=0): identical trajectory
    model2 = make_model(H, L, gen, "f32", vo.init_params(L, H, seed, gen), kld_weight=kw)
    opt2, sched2 = build_optimizer(cfg, model2, steps_per_epoch=total)
    it2 = iter(epss)
    orig2 = model2.fused_forward_backward
    model2.fused_forward_backward = lambda x, **k: orig2(x, **{**k, "eps": next(it2)})
    os.environ["VAE_ONE_CALL_STEP"] = "0"
    try:
        res2, _, _ = train_one_epoch(cfg, model2, opt2, sched2, model2.loss, loader, device="cuda", epoch=2)
    finally:
        del os.environ["VAE_ONE_CALL_STEP"]
    assert res2["loss"] == res["loss"]
    assert torch.equal(model2.flat_parameters(), model.flat_parameters())


def test_train_one_epoch_byte_stimuli_match_float_stimuli():
    """The loop fed uint8 pianorolls or bit planes (train.pack_bits; expanded to float32 on the device) walks the same trajectory,
    bit for bit, as the loop fed the float32 batches the reference's DataLoader hands over."""
    from argparse import Namespace
    from torch_vae_amd.train import build_optimizer, train_one_epoch
    H, L, B, steps = 32, 16, 6, 5
    cfg = Namespace(batch_size_per_gpu=B, world_size=1, lr_relative=0.01, weight_decay=0.0, optimizer="AdamW", scheduler="OneCycle",
                    epochs=1, log_wandb=False, print_interval=1000, log_interval=1000, freeze_encoder=False, global_rank=0)
    xs = [torch.from_numpy(vo.synth_pianoroll(B, H, 40 + s)) for s in range(steps)]
    assert all(bool(((x == 0) | (x == 1)).all()) for x in xs)
    out = []
    from torch_vae_amd.train import pack_bits
    for as_bytes in (False, True, "bits"):
        torch.manual_seed(0)
        model = make_model(H, L, False, "bf16", vo.init_params(L, H, 3, False), kld_weight=1.0)
        opt, sched = build_optimizer(cfg, model, steps_per_epoch=steps)
        epss = iter([torch.from_numpy(vo.counter_normal(B * L, 40 + s, 5).reshape(B, L)).float().cuda() for s in range(steps)])
        orig = model.fused_train_step
        model.fused_train_step = lambda o, x, _orig=orig, _it=epss, **k: _orig(o, x, **{**k, "eps": next(_it)})
        conv = {False: lambda t: t, True: lambda t: t.to(torch.uint8).pin_memory(), "bits": lambda t: pack_bits(t).pin_memory()}[as_bytes]
        loader = [(conv(x), torch.zeros(B, dtype=torch.long)) for x in xs]
        res, total_step, n_seen = train_one_epoch(cfg, model, opt, sched, model.loss, loader, device="cuda", epoch=2)
        assert total_step == steps and n_seen == steps * B
        out.append((res["loss"], model.flat_parameters().clone()))
    for o in out[1:]:
        assert out[0][0] == o[0]
        assert torch.equal(out[0][1], o[1])


def test_expand_stimuli_kernel_from_device_and_pinned_host_memory():
    """include/vae_step.h: vae_expand_stimuli - bytes and bit planes to float32, bit for bit the torch expression, from device memory
    and from pinned host memory read in place; pageable host memory, odd sizes and unknown kinds are refused."""
    from torch_vae_amd import _lib
    from torch_vae_amd.train import pack_bits, _expand_stimuli
    Lb = _lib.lib()
    st = torch.cuda.current_stream().cuda_stream
    g = torch.Generator().manual_seed(3)
    for (B, H) in ((3, 32), (5, 128), (256, 128)):
        cells = (torch.rand(B, 1, H, H, generator=g) < 0.3)
        want01 = cells.float().cuda()
        raw = torch.randint(0, 256, (B, 1, H, H), dtype=torch.uint8, generator=g)
        planes = pack_bits(cells)
        assert planes.shape == (B, 1, H, H // 8)
        for place in (lambda t: t.cuda(), lambda t: t.pin_memory()):
            for src, kind, want in ((cells.to(torch.uint8), 0, want01), (raw, 0, raw.float().cuda()), (planes, 1, want01)):
                s = place(src.contiguous())
                out = torch.full((B, 1, H, H), -7.0, device="cuda")
                _lib.check(Lb.vae_expand_stimuli(s.data_ptr(), kind, out.data_ptr(), B * H * H, st), "vae_expand_stimuli")
                torch.cuda.synchronize()
                assert torch.equal(out, want)
            # the host-side helper of the loop takes the same path (bool cells too)
            assert torch.equal(_expand_stimuli(place(planes), H, device="cuda"), want01)
            assert torch.equal(_expand_stimuli(place(cells), H, device="cuda"), want01)
        assert torch.equal(_expand_stimuli(planes, H, device="cuda"), want01)      # pageable host memory: copied first
    out = torch.zeros(64, device="cuda")
    pageable = torch.zeros(64, dtype=torch.uint8)
    assert Lb.vae_expand_stimuli(pageable.data_ptr(), 0, out.data_ptr(), 64, st) != 0
    dev = pageable.cuda()
    assert Lb.vae_expand_stimuli(dev.data_ptr(), 2, out.data_ptr(), 64, st) != 0
    assert Lb.vae_expand_stimuli(dev.data_ptr(), 0, out.data_ptr(), 60, st) != 0
    assert Lb.vae_expand_stimuli(None, 0, out.data_ptr(), 64, st) != 0
    assert Lb.vae_expand_stimuli(dev.data_ptr(), 0, out.data_ptr(), 0, st) == 0
    torch.cuda.synchronize()


def test_pipelined_and_simple_conv_kernels_agree():
    """The persistent/prefetched conv kernels (conv_pipe.cuh; wave-independent tiles and the 2x2 wave layout) against
    the one-tile-per-workgroup ones.  f32: exact arithmetic, only the summation order of the BatchNorm statistics
    differs.  bf16: the two families add the conv bias at different points of the f32 accumulation chain and sum
    statistics in a different order, so a few stored activations round the other way; through eight BatchNorm/
    LeakyReLU layers that decorrelates the roundings of the late layers completely (measured: half of the final
    layer's stored values differ by one bf16 ulp), and gradients then differ at the bf16 noise level (~10 %), the
    same level test_every_tensor_against_oracle allows against the oracle - so bf16 gets a direction check."""
    from torch_vae_amd import _lib
    cases = [(64, 16, 5, True, "bf16", {}), (32, 16, 9, False, "f32", {}), (128, 16, 3, True, "bf16", {}),
             (64, 16, 5, True, "f32", {"knob_lay22_min_nt": 2}), (128, 16, 3, True, "f32", {"knob_lay22_min_nt": 2}),
             (64, 16, 5, True, "f32", {"knob_wave_nt_max": 0, "knob_lay22_min_nt": 99})]
    for (H, L, B, gen, dtype, opts) in cases:
        p = perturbed_params(L, H, 8, gen)
        x = torch.from_numpy(vo.synth_pianoroll(B, H, 12)).cuda()
        eps = torch.from_numpy(vo.counter_normal(B * L, 12, 5).reshape(B, L)).float().cuda()
        res = []
        for use in (0, 1):
            model = make_model(H, L, gen, dtype, p)
            model._context(B)
            assert _lib.lib().vae_set_option(model._ctx.handle, b"use_pipelined", use) == 0
            for k, v in opts.items():
                assert _lib.lib().vae_set_option(model._ctx.handle, k.encode(), v) == 0
            out3, xhat = model.fused_forward_backward(x, eps=eps)
            res.append((out3.tolist(), xhat.clone(), flat_grad_dict(model)))
        np.testing.assert_allclose(res[1][0], res[0][0], rtol=1e-5 if dtype == "f32" else 3e-4)
        assert rel_l2(res[1][1].cpu().numpy(), res[0][1].cpu().numpy()) < (1e-5 if dtype == "f32" else 3e-3)
        for n in res[0][2]:
            if n in PRE_BN_BIAS:
                continue
            a_, b_ = res[1][2][n].astype(np.float64), res[0][2][n].astype(np.float64)
            if dtype == "f32":
                assert rel_l2(a_, b_) < 5e-3, (H, dtype, n)   # kink ties may flip with the summation order
            else:
                cos = float((a_ * b_).sum() / max(np.sqrt((a_ * a_).sum() * (b_ * b_).sum()), 1e-300))
                assert cos > 0.97 and 0.8 < np.sqrt((a_ * a_).sum() / max((b_ * b_).sum(), 1e-300)) < 1.25, (H, dtype, n, cos)


def test_decode_sample_evaluate_and_checkpoint(tmp_path):
    """N2-N4 of SURVEY.md 8f: decode(z)/sample(), the forward-only evaluate() consumer, checkpoint round trip."""
    from torch_vae_amd.evaluation import evaluate
    from torch_vae_amd.optim import FusedAdamW
    from torch_vae_amd.utils import safe_save_model
    H, L, B, gen = 32, 16, 8, False
    p = perturbed_params(L, H, 6, gen)
    m = make_model(H, L, gen, "f32", p)
    st = vo.init_bn_state()
    x = vo.synth_pianoroll(B, H, 2)
    eps = vo.counter_normal(B * L, 2, 5).reshape(B, L)
    # decode(z) in eval mode == decoder half of the oracle forward
    m.eval()
    c = vo.forward(p, x.astype(np.float64), eps, st, train=False)
    with torch.no_grad():
        xd = m.decode(torch.from_numpy(c["zlat"]).float().cuda())
    assert rel_l2(xd.cpu().numpy(), c["output"]) < 1e-4
    s = m.sample(5, "cuda")
    assert s.shape == (5, 1, H, H) and bool(((s >= 0) & (s <= 1)).all())
    # evaluate(): count / mse / mae like evaluation.py:96-100
    class DS(list):
        pass
    loader = [(torch.from_numpy(x[:4]), torch.zeros(4)), (torch.from_numpy(x[4:]), torch.zeros(4))]
    m.set_next_eps(torch.from_numpy(eps[:4]).float().cuda())
    res = evaluate(loader, m, "cuda", verbosity=0)
    assert res["count"] == B and res["cross-entropy"] == 0.0 and 0 < res["mae"] < 100 and 0 < res["mse"] < 100
    # checkpoint round trip in the reference's format (utils.py:311-351, train.py:444-460)
    m.train()
    opt = FusedAdamW([{"params": m.encoder.parameters()}, {"params": m.decoder.parameters()}], lr=1e-3, weight_decay=0.0)
    m.fused_forward_backward(torch.from_numpy(x).cuda(), eps=torch.from_numpy(eps).float().cuda())
    opt.step()
    path = str(tmp_path / "ckpt" / "model.pt")
    safe_save_model({"encoder": m.encoder, "decoder": m.decoder, "optimizer": opt}, path, epoch=3)
    ck = torch.load(path, weights_only=False)
    assert set(ck) >= {"encoder", "decoder", "optimizer", "epoch"} and "0.0.weight" in ck["encoder"]
    m2 = make_model(H, L, gen, "f32", p)
    m2.encoder.load_state_dict(ck["encoder"]); m2.decoder.load_state_dict(ck["decoder"])
    for (n, a), (_, b) in zip(m.encoder.state_dict().items(), m2.encoder.state_dict().items()):
        assert torch.equal(a, b), n
    opt2 = FusedAdamW([{"params": m2.encoder.parameters()}, {"params": m2.decoder.parameters()}], lr=1e-3, weight_decay=0.0)
    opt2.load_state_dict(ck["optimizer"])
    assert opt2._step == 1 and torch.equal(opt2._m, opt._m) and torch.equal(opt2._v, opt._v)
    # flat views survived load_state_dict: one more fused step moves both models identically
    for mm, oo in ((m, opt), (m2, opt2)):
        mm.fused_forward_backward(torch.from_numpy(x).cuda(), eps=torch.from_numpy(eps).float().cuda())
        oo.step()
    assert rel_l2(m2.flat_parameters().cpu().numpy(), m.flat_parameters().cpu().numpy()) < 1e-6


def test_c_abi_train_step_matches_python_path():
    """vae_train_step (forward + loss + backward + AdamW in one C call, train.py:634-659) against the same steps
    driven from Python (fused_forward_backward + FusedAdamW.step)."""
    import ctypes as C
    from torch_vae_amd import _lib
    from torch_vae_amd.optim import FusedAdamW
    H, L, B, gen, K = 32, 16, 8, False, 3
    p = perturbed_params(L, H, 11, gen)
    xs = [torch.from_numpy(vo.synth_pianoroll(B, H, 50 + i)).cuda() for i in range(K)]
    es = [torch.from_numpy(vo.counter_normal(B * L, 50 + i, 5).reshape(B, L)).float().cuda() for i in range(K)]
    lr, b1, b2, aeps, wd = 3e-3, 0.9, 0.999, 1e-8, 0.01
    # Python path
    m1 = make_model(H, L, gen, "f32", p)
    opt = FusedAdamW([{"params": list(m1.encoder.parameters())}, {"params": list(m1.decoder.parameters())}],
                     lr=lr, betas=(b1, b2), eps=aeps, weight_decay=wd)
    losses1 = []
    for i in range(K):
        out3, _ = m1.fused_forward_backward(xs[i], eps=es[i])
        opt.step()
        losses1.append(out3.tolist())
    # one C call per step on a second model's buffers
    m2 = make_model(H, L, gen, "f32", p)
    ctx = m2._context(B)
    flat, g = m2.flat_parameters(), m2.flat_grads()
    mom, var = torch.zeros_like(flat), torch.zeros_like(flat)
    opt._bind()
    offs = (C.c_int64 * 2)(*[r[0] for r in opt._ranges]); sizes = (C.c_int64 * 2)(*[r[1] for r in opt._ranges])
    lrs = (C.c_double * 2)(lr, lr); b1s = (C.c_double * 2)(b1, b1)
    xhat = torch.empty(B, 1, H, H, device="cuda"); mu = torch.empty(B, L, device="cuda"); lv = torch.empty_like(mu); z = torch.empty_like(mu)
    out3 = torch.empty(3, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    losses2 = []
    for i in range(K):
        _lib.check(_lib.lib().vae_train_step(ctx.handle, xs[i].data_ptr(), B, flat.data_ptr(), g.data_ptr(), mom.data_ptr(), var.data_ptr(),
                                             m2._bnflat.data_ptr(), m2._nbt.data_ptr(), es[i].data_ptr(), 0, 1.0, 2, offs, sizes, lrs, b1s,
                                             b2, aeps, wd, i + 1, xhat.data_ptr(), mu.data_ptr(), lv.data_ptr(), z.data_ptr(),
                                             out3.data_ptr(), st), "vae_train_step")
        losses2.append(out3.tolist())
    np.testing.assert_allclose(losses2, losses1, rtol=1e-6)
    np.testing.assert_array_equal(flat.cpu().numpy(), m1.flat_parameters().detach().cpu().numpy())
    np.testing.assert_array_equal(m2._bnflat.cpu().numpy(), m1._bnflat.cpu().numpy())
    assert m2._nbt.tolist() == m1._nbt.tolist() == [K] * 8


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_execution_options_do_not_change_results(dtype):
    """Side streams and the folded BatchNorm finalisation only change WHERE work runs: results are bit-identical.
    Wave-independent vs workgroup tiles change the order BatchNorm statistics are summed in: rounding-level agreement."""
    from torch_vae_amd import _lib
    H, L, B, gen = 64, 16, 6, True
    p = perturbed_params(L, H, 21, gen)
    x = torch.from_numpy(vo.synth_pianoroll(B, H, 31)).cuda()
    eps = torch.from_numpy(vo.counter_normal(B * L, 31, 5).reshape(B, L)).float().cuda()

    def run(opts):
        model = make_model(H, L, gen, dtype, p)
        model._context(B)
        for k, v in opts.items():
            assert _lib.lib().vae_set_option(model._ctx.handle, k.encode(), v) == 0
        res = []
        for _ in range(2):   # twice: the second pass also exercises re-use of the side streams / slabs
            out3, xhat = model.fused_forward_backward(x, eps=eps)
            res.append((out3.cpu().numpy().copy(), xhat.cpu().numpy().copy(), model.flat_grads().detach().cpu().numpy().copy(),
                        model._bnflat.cpu().numpy().copy()))
        return res

    base = run({})
    # knob_lean: which small launches stay off the caller's stream (7: also BatchNorm backward of block 0 inside its
    # weight-gradient kernel and the ELBO scalars on a side stream); knob_wgrad_mid8 is library-wide, reset below
    for opts in ({"use_side_stream": 0}, {"use_fused_bn": 0}, {"use_side_stream": 0, "use_fused_bn": 0}, {"knob_lean": 7},
                 {"knob_lean": 0}, {"knob_wgrad_mid8": 1}, {"knob_wgrad_mid8": 0}):
        got = run(opts)
        for a, b in zip(got, base):
            for u, v in zip(a, b):
                np.testing.assert_array_equal(u, v, err_msg=str(opts))
    got = run({"knob_wave_nt_max": 0})
    for a, b in zip(got, base):
        np.testing.assert_allclose(a[0], b[0], rtol=1e-6 if dtype == "f32" else 3e-4)
        assert rel_l2(a[1], b[1]) < (1e-5 if dtype == "f32" else 3e-3)
        assert rel_l2(a[2], b[2]) < (5e-3 if dtype == "f32" else 5e-2)


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_full_size_properties(dtype):
    """BASELINE workload size (128x128, latent 16, batch 256 - too large for the numpy oracle in a test): properties
    that hold at any size, checked with independent torch reductions on the device.
      * reconstruction term == BCE(xhat, x) from the returned xhat; KL term from the returned mu/log_var;
      * BatchNorm running statistics of the first and last layer == statistics of the stored activations;
      * the backward is linear in the upstream gradient: loss*2 gives exactly 2x gradients (powers of two are exact
        in f32 and bf16), and a second pass over the same batch reproduces the first bit for bit (determinism of the
        split-K slabs / side streams);
      * permuting the batch leaves the ELBO and the gradients unchanged up to summation order;
      * one fused AdamW step == torch.optim.AdamW on the same gradients."""
    import torch.nn.functional as F
    from torch_vae_amd import _lib
    from torch_vae_amd.optim import FusedAdamW
    H, L, B, gen = 128, 16, 256, True
    p = perturbed_params(L, H, 5, gen)
    x = torch.from_numpy(vo.synth_pianoroll(B, H, 77)).cuda()
    eps = torch.from_numpy(vo.counter_normal(B * L, 77, 5).reshape(B, L)).float().cuda()
    tol = 1e-5 if dtype == "f32" else 2e-4
    model = make_model(H, L, gen, dtype, p, kld_weight=4.0)
    out3, xhat = model.fused_forward_backward(x, eps=eps)
    g1 = model.flat_grads().detach().clone()
    mu, lv = model._last["mu"], model._last["lv"]
    assert float(xhat.min()) >= 0.0 and float(xhat.max()) <= 1.0
    bce = F.binary_cross_entropy(xhat.double(), x.double()).item()
    kld = (-0.5 * (1 + lv.double() - mu.double() ** 2 - lv.double().exp()).sum(1)).mean().item()
    np.testing.assert_allclose(out3[1].item(), bce, rtol=tol)
    np.testing.assert_allclose(-out3[2].item(), kld, rtol=tol)          # the reference's kld_loss carries the flipped sign
    np.testing.assert_allclose(out3[0].item(), bce + 4.0 * kld, rtol=tol)
    # BatchNorm statistics of the stored tensors (f64 atomics over 4M / 4M pixels)
    sd = model.state_dict()
    for which, key, C, S in ((0, "encoder.0.1", 32, H // 2), (7, "final_layer.1", 32, H)):
        n = B * C * S * S
        y = torch.empty(n, device="cuda")
        _lib.check(_lib.lib().vae_debug_tensor(model._ctx.handle, which, y.data_ptr(), n, torch.cuda.current_stream().cuda_stream), "dbg")
        y = y.view(B, C, S, S).double()
        mean, var = y.mean((0, 2, 3)), y.var((0, 2, 3), unbiased=True)
        np.testing.assert_allclose(sd[key + ".running_mean"].cpu().numpy(), (0.1 * mean).cpu().numpy(), rtol=1e-4, atol=1e-6)
        np.testing.assert_allclose(sd[key + ".running_var"].cpu().numpy(), (0.9 + 0.1 * var).cpu().numpy(), rtol=1e-4)
    # determinism and linearity of the backward
    out3b, _ = model.fused_forward_backward(x, eps=eps)
    assert torch.equal(model.flat_grads(), g1) and torch.equal(out3b, out3)
    m2 = make_model(H, L, gen, dtype, p, kld_weight=4.0)
    m2.set_next_eps(eps)
    (2.0 * m2.loss(m2(x))["loss"]).backward()
    if dtype != "f32" and H == 128:
        # (the one-pass step takes the row-streaming output-conv kernel, the autograd path the tiled kernels: their BatchNorm-
        #  backward statistics differ in the last f32 bits, which moves a few 16-bit roundings upstream)
        lin = rel_l2(m2.flat_grads().cpu().numpy(), 2.0 * g1.cpu().numpy())
        report(test="full_size_properties_linearity", dtype=dtype, rel_l2=lin)
        assert lin < {"bf16": 5e-4, "f16": 2e-4}[dtype], lin   # (measured 2.4e-5 at bf16)
    else:
        assert torch.equal(m2.flat_grads(), 2.0 * g1)
    # batch permutation
    perm = torch.randperm(B, generator=torch.Generator().manual_seed(3)).cuda()
    m3 = make_model(H, L, gen, dtype, p, kld_weight=4.0)
    out3p, _ = m3.fused_forward_backward(x[perm].contiguous(), eps=eps[perm].contiguous())
    np.testing.assert_allclose(out3p.cpu().numpy(), out3.cpu().numpy(), rtol=1e-5 if dtype == "f32" else 5e-4)
    gp = m3.flat_grads().double()
    if dtype == "f32":
        assert rel_l2(gp.cpu().numpy(), g1.double().cpu().numpy()) < 5e-3
    else:
        cos = float((gp * g1.double()).sum() / (gp.norm() * g1.double().norm()))
        assert cos > 0.97
    # fused AdamW == torch.optim.AdamW
    ref_p = model.flat_parameters().detach().clone().requires_grad_(True)
    ref_p.grad = g1.clone()
    ref_opt = torch.optim.AdamW([ref_p], lr=2e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01)
    opt = FusedAdamW([{"params": list(model.encoder.parameters())}, {"params": list(model.decoder.parameters())}],
                     lr=2e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01)
    opt._bind()
    before = model.flat_parameters().detach().clone()
    opt.step(); ref_opt.step()
    after = model.flat_parameters().detach()
    for (o, n) in opt._ranges:
        np.testing.assert_allclose(after[o:o + n].cpu().numpy(), ref_p.detach()[o:o + n].cpu().numpy(), rtol=1e-6, atol=1e-9)
    touched = torch.zeros_like(before, dtype=torch.bool)
    for (o, n) in opt._ranges:
        touched[o:o + n] = True
    assert torch.equal(after[~touched], before[~touched])     # fc_mu / fc_var / decoder_input / final_layer are not optimised (train.py:228)


def test_split_backward_and_overlapped_allreduce():
    """vae_backward_part(1) + (2) == vae_backward (bit for bit), the decoder gradients are final when the hook runs,
    and train.fused_step under an initialised (single-rank, RCCL) process group takes the same step as without one."""
    import os as _os
    import torch.distributed as dist
    from torch_vae_amd.optim import FusedAdamW
    from torch_vae_amd.train import fused_step
    H, L, B, gen = 64, 16, 6, True
    p = perturbed_params(L, H, 33, gen)
    x = torch.from_numpy(vo.synth_pianoroll(B, H, 9)).cuda()
    eps = torch.from_numpy(vo.counter_normal(B * L, 9, 5).reshape(B, L)).float().cuda()
    for dtype in ("f32", "bf16"):
        m0 = make_model(H, L, gen, dtype, p)
        m0.fused_forward_backward(x, eps=eps)
        g0 = m0.flat_grads().detach().clone()
        m1 = make_model(H, L, gen, dtype, p)
        seen = {}

        def hook():
            off, n = m1.group_range("decoder")
            seen["decoder"] = m1.flat_grads()[off:off + n].detach().clone()
        m1.fused_forward_backward(x, eps=eps, on_decoder_grads=hook)
        assert torch.equal(m1.flat_grads(), g0)
        off, n = m1.group_range("decoder")
        assert torch.equal(seen["decoder"], g0[off:off + n])
    # same optimiser step with and without a process group
    def one_step(use_dist):
        m = make_model(H, L, gen, "f32", p)
        opt = FusedAdamW([{"params": list(m.encoder.parameters())}, {"params": list(m.decoder.parameters())}], lr=1e-3)
        out3, _ = fused_step(m, opt, x, eps=eps)
        return out3.cpu().numpy(), m.flat_parameters().detach().cpu().numpy().copy()
    ref = one_step(False)
    _os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); _os.environ.setdefault("MASTER_PORT", "29571")
    _os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        got = one_step(True)
    finally:
        dist.destroy_process_group()
    np.testing.assert_array_equal(got[0], ref[0])
    np.testing.assert_array_equal(got[1], ref[1])


def test_bench_two_rank_control_flow():
    """bench.py under torch.distributed.run with two ranks finishes and rank 0 prints ONE JSON line: every rank has to
    take part in every step that contains the gradient all-reduce, including the profiled steps after the timed
    region.  Rehearsal only: both ranks share this box's GPU, so the exchange goes through gloo instead of RCCL."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29587", os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
           "--size", "32", "--batch", "32", "--backend", "gloo", "--no-cpu-baseline"]
    r = subprocess.run(cmd, cwd=root, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    # (two ranks time-slice one GPU here, so a launch can take milliseconds and its GB/s round to 0.0: check the inputs)
    assert out["n_gpus"] == 2 and out["config"]["global_batch"] == 64
    assert out["roofline"]["algorithmic_bytes_per_launch"] > 0 and out["roofline"]["avg_launch_us"] > 0


# ---------------------------------------------------------------------------------------------------------------------
# round 2: every BASELINE.json configuration on the HIP path, the 16-bit modes gated at what they measure, the reference's
# edge cases on the HIP kernels, fallbacks forced, N2/N3 pinned to reference-written fixtures
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", ["G_h256_l64_b2", "G_h128_l128_b2_k1", "G_h128_l128_b2_k16", "G_h128_l128_b2", "G_h128_l16_b2", "R_l10_b8"])
@pytest.mark.parametrize("dtype", ["bf16", "f16"])
def test_16bit_step0_against_reference_fixture(name, dtype):
    """The 16-bit storage modes on the BASELINE shapes (configs[2]: 256x256 latent 64; configs[4]: 128x128 latent 128 at
    beta 1 / 4 / 16) against the REFERENCE's fp64 outputs: measured gap reported, bounded by ELBO_TOL / GRAD_TOL."""
    H, L, B, steps, total, kw, gen, seed = CASES[name]
    gold = np.load(os.path.join(GOLD, f"{name}_f64.npz"))
    model = make_model(H, L, gen, dtype, vo.init_params(L, H, seed, gen), kld_weight=kw)
    x, eps = case_inputs(name, 0)
    out3, xhat = model.fused_forward_backward(torch.from_numpy(x).cuda(), eps=torch.from_numpy(eps).float().cuda())
    got = np.array(out3.tolist())
    elbo_gap = float(np.abs(got / gold["losses"][0] - 1).max())
    g = flat_grad_dict(model)
    norm_gap = {n: abs(float(np.sqrt((v.astype(np.float64) ** 2).sum())) / float(gold["gradnorm/" + n]) - 1)
                for n, v in g.items() if n not in PRE_BN_BIAS}
    worst = max(norm_gap, key=norm_gap.get)
    report(test="fixture16", case=name, dtype=dtype, elbo_rel=elbo_gap, gradnorm_rel_max=norm_gap[worst], worst=worst)
    assert elbo_gap < ELBO_TOL[dtype]
    assert rel_l2(model._last["mu"].cpu().numpy(), gold["mu"]) < 1.5 * FWD_TOL[dtype]
    xh = xhat.double().cpu().numpy()
    assert abs(xh.sum() / float(gold["output_sum"]) - 1) < FWD_TOL[dtype]
    for n, gap in norm_gap.items():
        assert gap < GRAD_TOL_FIXTURE[dtype], (n, gap)


FULL = [
    # dtype, H, L, B, kld_weight         BASELINE.json configs[1] (bf16; f32 = the 1e-4 parity mode of the same workload),
    ("f32", 128, 16, 256, 4.0),        # configs[2] (256x256, latent 64, batch 512, bf16 + f32 loss),
    ("bf16", 128, 16, 256, 4.0),       # configs[4] (128x128, latent 128, 512 per GPU, f16 storage + f32 KL accumulate, beta 16)
    ("bf16", 256, 64, 512, 1.0),
    ("f16", 128, 128, 512, 16.0),
]


@pytest.mark.parametrize("cfg", FULL, ids=[f"{d}-{h}x{h}-L{l}-B{b}" for d, h, l, b, _ in FULL])
def test_full_size_baseline_configs(cfg):
    """Every single-GPU BASELINE.json configuration at FULL size on the HIP path (too large for the numpy oracle):
    size-independent properties checked with independent torch reductions on the device.
      * reconstruction term == BCE(xhat, x) from the returned xhat; KL term from the returned mu/log_var (f64);
      * the backward is linear in the upstream gradient (loss*2 -> 2x gradients; exact for f32/bf16, rounding-level for
        f16 whose stored gradients are rescaled), a second pass over the same batch reproduces the first bit for bit;
      * the 16-bit ELBO against the f32 mode on the same batch (the 1e-4 mode, itself pinned to the reference at small
        sizes): gap reported and bounded."""
    import torch.nn.functional as F
    dtype, H, L, B, kw = cfg
    p = perturbed_params(L, H, 5, True)
    x = torch.from_numpy(vo.synth_pianoroll(B, H, 77)).cuda()
    eps = torch.from_numpy(vo.counter_normal(B * L, 77, 5).reshape(B, L)).float().cuda()
    tol = {"f32": 1e-5, "bf16": 2e-4, "f16": 2e-4}[dtype]
    model = make_model(H, L, True, dtype, p, kld_weight=kw)
    out3, xhat = model.fused_forward_backward(x, eps=eps)
    g1 = model.flat_grads().detach().clone()
    assert bool(torch.isfinite(g1).all()) and bool(torch.isfinite(out3).all())
    mu, lv = model._last["mu"], model._last["lv"]
    bce = F.binary_cross_entropy(xhat.double(), x.double()).item()
    kld = (-0.5 * (1 + lv.double() - mu.double() ** 2 - lv.double().exp()).sum(1)).mean().item()
    np.testing.assert_allclose(out3[1].item(), bce, rtol=tol)
    np.testing.assert_allclose(-out3[2].item(), kld, rtol=tol)
    np.testing.assert_allclose(out3[0].item(), bce + kw * kld, rtol=tol)
    out3b, _ = model.fused_forward_backward(x, eps=eps)
    assert torch.equal(model.flat_grads(), g1) and torch.equal(out3b, out3)
    del model
    m2 = make_model(H, L, True, dtype, p, kld_weight=kw)
    m2.materialize_pre_latents = False
    m2.set_next_eps(eps)
    (2.0 * m2.loss(m2(x))["loss"]).backward()
    if dtype == "f16":
        assert rel_l2(m2.flat_grads().cpu().numpy(), 2.0 * g1.cpu().numpy()) < 2e-4
    elif dtype == "bf16" and H == 128:    # (row-streaming vs tiled output-conv kernel: statistics summed in another order)
        lin = rel_l2(m2.flat_grads().cpu().numpy(), 2.0 * g1.cpu().numpy())
        report(test="full_size_baseline_linearity", dtype=dtype, cfg=[H, L, B], rel_l2=lin)
        assert lin < 5e-4, lin   # (measured 2.4e-5)
    else:
        assert torch.equal(m2.flat_grads(), 2.0 * g1)
    del m2
    if dtype != "f32" and (H, L, B) != (256, 64, 512):   # (f32 at 256x256 / 512 takes the slow 64-bit-offset kernels: covered at B=2)
        m3 = make_model(H, L, True, "f32", p, kld_weight=kw)
        ref3, _ = m3.fused_forward_backward(x, eps=eps)
        gap = float((out3 / ref3 - 1).abs().max())
        gref = m3.flat_grads().double()
        ggap = float((g1.double() - gref).norm() / gref.norm())
        report(test="full_size", cfg=list(cfg), elbo_rel_vs_f32=gap, flat_grad_rel_l2_vs_f32=ggap)
        assert gap < ELBO_TOL[dtype]
        assert ggap < GRAD_TOL_FULL[dtype]


# BASELINE.json single-GPU configurations at FULL size against the torch-CPU oracle (oracle/torch_cpu_step.py, pinned to the
# reference by tests/test_oracle.py): configs[1] 128x128 L16 B256, configs[2] 256x256 L64 B512, configs[4] 128x128 L128 B512 beta 16.
# The host side of configs[2] at its full batch of 512 takes ~170 s (and two layer-local cases below ~110 s together): the default
# run keeps the suite near seven minutes by taking configs[2] (and the two layer-local cases) at a reduced batch - same models, same
# kernels, several tiles / bands per workgroup - and runs configs[1] and configs[4] exactly; VAE_FULL_TESTS=1 runs every case at the
# exact BASELINE size (it passes: profiles/r03_gpu_tests_full.txt).
FULL_TESTS = os.environ.get("VAE_FULL_TESTS", "0") == "1"
FULL_ORACLE = [(128, 16, 256, 1.0), (128, 128, 512, 16.0), (256, 64, 512 if FULL_TESTS else 64, 1.0)]


@pytest.mark.parametrize("cfg", FULL_ORACLE, ids=[f"{h}x{h}-L{l}-B{b}-k{int(k)}" for h, l, b, k in FULL_ORACLE])
def test_full_size_against_cpu_oracle(cfg):
    """The f32 kernel mode at the full BASELINE sizes against the oracle ON THE SAME weights / batch / eps: ELBO scalars and
    mu / log_var within 1e-4 (north_star), the whole flat gradient within 5e-3 rel-L2 (LeakyReLU kink ties) and
    every parameter tensor's gradient within 2e-2.  Then the 16-bit modes against that same oracle result, their gaps
    reported and bounded (the indexing of every layer at full batch is covered: a deterministic, linear large-batch bug
    in a middle layer would pass the size-independent property tests above, not this one)."""
    from oracle.torch_cpu_step import TorchCpuStep
    import torch.nn.functional as F
    H, L, B, kw = cfg
    p = perturbed_params(L, H, 5, True)
    x_np = vo.synth_pianoroll(B, H, 77)
    eps_np = vo.counter_normal(B * L, 77, 5).reshape(B, L).astype(np.float32)
    # oracle: forward, ELBO, backward (train.py:634-650) on the host
    torch.set_num_threads(max(1, os.cpu_count() or 1))
    st = TorchCpuStep(p, kld_weight=kw, batch=B, total_steps=4)
    xc = torch.from_numpy(x_np)
    xhat_c, mu_c, lv_c, _ = st.forward(xc, torch.from_numpy(eps_np))
    recon = F.binary_cross_entropy(xhat_c, xc)
    kld = -0.5 * torch.mean(torch.sum(1 + lv_c - mu_c ** 2 - torch.exp(lv_c), dim=-1))
    (recon + kw * kld).backward()
    want3 = np.array([float(recon + kw * kld), float(recon), float(-kld)])
    gref = {k: v.grad.detach().numpy().ravel().astype(np.float64) for k, v in st.p.items()}
    mu_c, lv_c = mu_c.detach().numpy(), lv_c.detach().numpy()
    del xhat_c, recon, kld, st
    names = [n for n in gref if n not in PRE_BN_BIAS]   # (conv biases in front of a train-mode BatchNorm: analytically zero)
    ref_flat = np.concatenate([gref[n] for n in names])
    x = torch.from_numpy(x_np).cuda(); eps = torch.from_numpy(eps_np).cuda()
    for dtype in ("f32", "bf16", "f16"):
        if dtype == "f32" and H == 256:
            continue   # (f32 storage at 256x256 / 512 takes the slow 64-bit-offset kernels: the f32 mode is covered at the 128x128 sizes)
        model = make_model(H, L, True, dtype, p, kld_weight=kw)
        out3, _ = model.fused_forward_backward(x, eps=eps)
        got3 = out3.cpu().numpy().astype(np.float64)
        g = flat_grad_dict(model)
        got_flat = np.concatenate([g[n].astype(np.float64) for n in names])
        elbo_gap = float(np.max(np.abs(got3 / want3 - 1)))
        mu_gap = rel_l2(model._last["mu"].cpu().numpy(), mu_c); lv_gap = rel_l2(model._last["lv"].cpu().numpy(), lv_c)
        flat_gap = rel_l2(got_flat, ref_flat)
        worst = max(((rel_l2(g[n], gref[n]), n) for n in names), key=lambda t: t[0])
        report(test="full_size_vs_cpu_oracle", cfg=list(cfg), dtype=dtype, elbo_rel=elbo_gap, mu_rel_l2=mu_gap, log_var_rel_l2=lv_gap,
               flat_grad_rel_l2=flat_gap, worst_tensor=worst[1], worst_tensor_rel_l2=worst[0])
        if dtype == "f32":
            assert elbo_gap < 1e-4 and mu_gap < 1e-4 and lv_gap < 1e-4, (cfg, elbo_gap, mu_gap, lv_gap)
            assert flat_gap < 5e-3 and worst[0] < 2e-2, (cfg, flat_gap, worst)
        else:
            assert elbo_gap < ELBO_TOL[dtype] and mu_gap < FWD_TOL[dtype] and lv_gap < FWD_TOL[dtype], (cfg, dtype, elbo_gap, mu_gap, lv_gap)
            # (per tensor: the wide small-tensor gate - a 32-element BatchNorm bias gradient measures 0.10 in bf16 here, LeakyReLU kink
            #  flips of the stored pre-activations; the whole flat gradient is gated tightly)
            assert flat_gap < GRAD_TOL_FULL[dtype] and worst[0] < GRAD_TOL[dtype], (cfg, dtype, flat_gap, worst)
        del model


def test_bce_edges_and_saturating_logits_on_hip():
    """ATen's BCE conventions (log clamp -100, gradient clamp 1e-12; reference models.py:208) on the HIP kernels:
    the reference-generated edge fixture through vae_elbo_generic, and saturating logits through the fused
    output-conv + sigmoid + BCE kernels of all three storage modes against the oracle."""
    from torch_vae_amd import _lib
    gold = np.load(os.path.join(GOLD, "bce_edges.npz"))
    n = gold["x"].size
    xh = torch.from_numpy(gold["x"]).cuda(); tg = torch.from_numpy(gold["t"]).cuda()
    mu = torch.zeros(1, 1, device="cuda"); lv = torch.zeros(1, 1, device="cuda")
    out3 = torch.empty(3, device="cuda"); gx = torch.empty(n, device="cuda"); gm = torch.empty(1, 1, device="cuda"); gl = torch.empty(1, 1, device="cuda")
    _lib.check(_lib.lib().vae_elbo_generic(xh.data_ptr(), tg.data_ptr(), mu.data_ptr(), lv.data_ptr(), n, 1, 1, 1.0, out3.data_ptr(),
                                          gx.data_ptr(), gm.data_ptr(), gl.data_ptr(), torch.cuda.current_stream().cuda_stream), "elbo")
    np.testing.assert_allclose(out3[1].item(), gold["per_elem"].astype(np.float64).mean(), rtol=1e-6)
    np.testing.assert_allclose(gx.cpu().numpy() * n, gold["grad"], rtol=1e-6)     # (x - t) / max(x (1 - x), 1e-12)
    assert out3[2].item() == 0.0
    # saturating logits: an output bias of +-150 drives sigmoid to exactly 1 / 0 in f32, so every pixel hits a clamp
    H, L, B = 32, 16, 4
    x = vo.synth_pianoroll(B, H, 3)
    eps = vo.counter_normal(B * L, 3, 5).reshape(B, L)
    # (bias 30 saturates f32 too: sigmoid(30) rounds to exactly 1.0f, as in the reference's f32 arithmetic; bias 8 does not)
    for bias in (150.0, -150.0, 30.0, 8.0):
        p = perturbed_params(L, H, 4, False)
        p["final_layer.3.bias"] = np.array([bias])
        c = vo.forward({k: v.astype(np.float32) for k, v in p.items()}, x.astype(np.float32), eps.astype(np.float32), None, train=True)
        want = vo.loss(c)     # the oracle in f32, the reference's arithmetic: saturation happens where f32 saturates
        for dtype in ("f32", "bf16", "f16"):
            m = make_model(H, L, False, dtype, p)
            out3, xhat = m.fused_forward_backward(torch.from_numpy(x).cuda(), eps=torch.from_numpy(eps).float().cuda())
            assert bool(torch.isfinite(m.flat_grads()).all())
            if abs(bias) > 20:
                # the loss is count(mismatching pixels) * 100 / N exactly, as ATen's log clamp gives it
                np.testing.assert_allclose(out3[1].item(), float(want["reconstruction_loss"]), rtol=1e-5)
                assert float(xhat.max()) == float(xhat.min()) == (1.0 if bias > 0 else 0.0)
                assert float(want["reconstruction_loss"]) > 5.0
            else:
                np.testing.assert_allclose(out3[1].item(), float(want["reconstruction_loss"]), rtol=ELBO_TOL[dtype] if dtype != "f32" else 2e-4)


def test_wgrad_64bit_offset_fallback_agrees():
    """The prefetching weight-gradient kernels index with 32-bit byte offsets; tensors of 4 GiB or more take the
    synchronous kernel with 64-bit offsets.  Forced here (knob_wgrad_force_simple) at a small size: same operands and the
    same MFMA products, only the split of K differs."""
    from torch_vae_amd import _lib
    H, L, B, gen = 64, 16, 6, True
    p = perturbed_params(L, H, 5, gen)
    x = torch.from_numpy(vo.synth_pianoroll(B, H, 9)).cuda()
    eps = torch.from_numpy(vo.counter_normal(B * L, 9, 5).reshape(B, L)).float().cuda()
    for dtype in ("bf16", "f16"):
        grads = []
        for force in (0, 1):
            model = make_model(H, L, gen, dtype, p)
            model._context(B)
            assert _lib.lib().vae_set_option(model._ctx.handle, b"knob_wgrad_force_simple", force) == 0
            model.fused_forward_backward(x, eps=eps)
            grads.append(model.flat_grads().clone())
        assert rel_l2(grads[1].cpu().numpy(), grads[0].cpu().numpy()) < 1e-5


def test_backward_of_a_stale_forward_raises():
    """The context keeps ONE forward's activations: backward of an older graph must fail loudly, not silently use the
    newer batch (the reference's autograd would handle it; here it is an explicit error)."""
    H, L, B = 32, 16, 4
    m = make_model(H, L, False, "f32", perturbed_params(L, H, 2, False))
    x1 = torch.from_numpy(vo.synth_pianoroll(B, H, 1)).cuda(); x2 = torch.from_numpy(vo.synth_pianoroll(B, H, 2)).cuda()
    l1 = m.loss(m(x1))["loss"]
    l2 = m.loss(m(x2))["loss"]
    with pytest.raises(RuntimeError, match="no longer the model's last"):
        l1.backward()
    l2.backward()
    assert bool(torch.isfinite(m.flat_grads()).all())


def test_reference_checkpoint_loads_and_resumes(tmp_path):
    """N2 pinned: a checkpoint WRITTEN BY THE REFERENCE's utils.safe_save_model (tests/golden/ckpt_R_b4_k1_f32.pt, made by
    make_golden.py after the 3 steps of R_b4_k1) loads as train.py:320-329 does, and the next two steps reproduce the
    losses the reference's own loop produced from that state; a checkpoint written by this package has the same keys
    and tensor shapes, and round-trips."""
    from argparse import Namespace
    from torch_vae_amd.train import build_optimizer
    from torch_vae_amd.utils import checkpoint_modules, load_checkpoint, safe_save_model
    name = "R_b4_k1"
    H, L, B, steps, total, kw, gen, seed = CASES[name]
    nxt = np.load(os.path.join(GOLD, "ckpt_R_b4_k1_f32_next.npz"))
    ck = torch.load(os.path.join(GOLD, "ckpt_R_b4_k1_f32.pt"), weights_only=False, map_location="cpu")
    assert ck["epoch"] == 2 and ck["total_step"] == steps and ck["n_samples_seen"] == steps * B
    # fc_mu / fc_var / decoder_input / final_layer are not in the reference's checkpoints (train.py:444-460) and are never
    # optimised (train.py:210-225): they keep their initial values, which both sides regenerate from the counter generator
    model = make_model(H, L, gen, "f32", vo.init_params(L, H, seed, gen), kld_weight=kw)
    cfg = Namespace(batch_size_per_gpu=B, world_size=1, lr_relative=0.01, weight_decay=0.0, optimizer="AdamW", scheduler="OneCycle",
                    epochs=1, freeze_encoder=False)
    opt, sched = build_optimizer(cfg, model, steps_per_epoch=total)
    state = load_checkpoint(model, opt, sched, ck)
    assert state["total_step"] == steps and state["n_samples_seen"] == steps * B
    got = []
    for s in range(steps, steps + 2):
        x, eps = case_inputs(name, s)
        out3, _ = model.fused_forward_backward(torch.from_numpy(x).cuda(), eps=torch.from_numpy(eps).float().cuda())
        opt.step(); sched.step()
        got.append(out3.tolist())
    np.testing.assert_allclose(np.array(got), nxt["resumed_losses"], rtol=3e-4)
    # the reverse direction: same keys, same tensor shapes, same optimizer / scheduler state layout
    path = str(tmp_path / "run" / "checkpoint_latest.pt")
    safe_save_model(checkpoint_modules(model, opt, sched), path, config=cfg, epoch=3, total_step=steps + 2, n_samples_seen=(steps + 2) * B)
    mine = torch.load(path, weights_only=False, map_location="cpu")
    assert set(mine) >= set(ck) - {"best_epoch"}
    for mod in ("encoder", "decoder"):
        assert list(mine[mod]) == list(ck[mod])
        assert all(mine[mod][k].shape == ck[mod][k].shape and mine[mod][k].dtype == ck[mod][k].dtype for k in ck[mod])
    assert set(mine["optimizer"]) == set(ck["optimizer"]) and len(mine["optimizer"]["state"]) == len(ck["optimizer"]["state"])
    for i, st in ck["optimizer"]["state"].items():
        assert set(mine["optimizer"]["state"][i]) == set(st)
        assert all(mine["optimizer"]["state"][i][k].shape == st[k].shape for k in ("exp_avg", "exp_avg_sq"))
    for gm, gr in zip(mine["optimizer"]["param_groups"], ck["optimizer"]["param_groups"]):
        assert set(gm) >= set(gr) and gm["params"] == gr["params"]
    assert set(mine["scheduler"]) == set(ck["scheduler"])
    # all-modules flag (the reference loses fc_mu / fc_var / decoder_input / final_layer on resume)
    full = checkpoint_modules(model, opt, sched, save_all_modules=True)
    assert {"fc_mu", "fc_var", "decoder_input", "final_layer"} <= set(full)
    # round trip into a fresh model continues identically
    m2 = make_model(H, L, gen, "f32", vo.init_params(L, H, seed, gen), kld_weight=kw)
    opt2, sched2 = build_optimizer(cfg, m2, steps_per_epoch=total)
    load_checkpoint(m2, opt2, sched2, mine)
    x, eps = case_inputs(name, steps + 2)
    for mm, oo in ((model, opt), (m2, opt2)):
        mm.fused_forward_backward(torch.from_numpy(x).cuda(), eps=torch.from_numpy(eps).float().cuda())
        oo.step()
    assert rel_l2(m2.flat_parameters().cpu().numpy(), model.flat_parameters().cpu().numpy()) < 1e-6


def test_evaluate_matches_reference_values():
    """N3 pinned: count / cross-entropy / mse / mae the REFERENCE's evaluation.evaluate produced (tests/golden/evaluate_R.npz)
    on a two-batch loader whose dataset is one sample short of the batches (the padding trim, evaluation.py:88-95)."""
    from torch_vae_amd.evaluation import evaluate
    gold = np.load(os.path.join(GOLD, "evaluate_R.npz"))
    H, L, B, seed, n_samples = 32, 16, 4, 21, 7
    m = make_model(H, L, False, "f32", vo.init_params(L, H, seed, False))

    class Loader(list):
        pass

    loader = Loader((torch.from_numpy(vo.synth_pianoroll(B, H, seed * 1000 + i)), torch.zeros(B, dtype=torch.long)) for i in range(2))
    loader.dataset = range(n_samples)
    epss = [torch.from_numpy(vo.counter_normal(B * L, seed * 1000 + i, 5).reshape(B, L)).float().cuda() for i in range(2)]
    fwd = m.forward

    def forward_with_eps(x):
        m.set_next_eps(epss.pop(0)[: x.shape[0]])
        return fwd(x)

    m.forward = forward_with_eps
    res = evaluate(loader, m, "cuda", verbosity=0)
    assert res["count"] == int(gold["count"]) == n_samples
    assert res["cross-entropy"] == float(gold["cross-entropy"]) == 0.0
    np.testing.assert_allclose(res["mse"], float(gold["mse"]), rtol=1e-4)
    np.testing.assert_allclose(res["mae"], float(gold["mae"]), rtol=1e-4)


def test_data_parallel_product_path_two_ranks():
    """train.fused_step / build_optimizer with world_size 2 (two ranks sharing this GPU over gloo) against the oracle's
    R-replica simulation: initial broadcast from deliberately different replicas, identical parameters on both ranks
    after every step, mean gradients in .grad, per-rank reparameterisation noise, lr scaling (tests/dp_product_worker.py)."""
    import subprocess
    import sys
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29547", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29547", os.path.join(ROOT, "tests", "dp_product_worker.py"), ROOT]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + "\n".join(ln for ln in r.stderr.splitlines() if "rank" in ln or "Error" in ln or "assert" in ln)[-6000:]
    assert r.stdout.count("DP_PRODUCT_OK") == 2


def test_library_allreduce_single_rank_rccl():
    """vae_comm_init / vae_allreduce_grads / vae_broadcast_state on a real RCCL communicator (one rank: the only size a
    one-GPU box can form): the bucketed exchange (vae_train_step_fused exchange = 2: decoder bucket on the communication stream
    under the encoder backward, each group's AdamW behind its own bucket's event) and the in-line exchange (exchange = 1) both
    leave two consecutive steps unchanged bit for bit, and train.fused_step routes through the library when asked to."""
    import ctypes as C
    import torch.distributed as dist
    from torch_vae_amd import _lib
    from torch_vae_amd.optim import FusedAdamW
    from torch_vae_amd.train import enable_library_allreduce, fused_step
    H, L, B, gen = 64, 16, 6, True
    p = perturbed_params(L, H, 33, gen)
    x = torch.from_numpy(vo.synth_pianoroll(B, H, 9)).cuda()
    eps = torch.from_numpy(vo.counter_normal(B * L, 9, 5).reshape(B, L)).float().cuda()

    def one_step(mode):
        m = make_model(H, L, gen, "f32", p)
        opt = FusedAdamW([{"params": list(m.encoder.parameters())}, {"params": list(m.decoder.parameters())}], lr=1e-3)
        if mode is not None:
            assert enable_library_allreduce(m)
        out3, _ = fused_step(m, opt, x, eps=eps, overlap=mode)
        out3 = out3.clone()
        fused_step(m, opt, x, eps=eps, overlap=mode)          # a second step: the update of the first one went through the bucketed / in-line AdamW launches
        if mode is not None:
            assert m.library_comm_world() == 1
            st = torch.cuda.current_stream().cuda_stream
            before = m.flat_parameters().detach().clone()
            _lib.check(_lib.lib().vae_broadcast_state(m._ctx.handle, m.flat_parameters().data_ptr(), m._bnflat.data_ptr(), m._nbt.data_ptr(), 0, st), "bcast")
            assert torch.equal(before, m.flat_parameters())
        return out3.cpu().numpy(), m.flat_parameters().detach().cpu().numpy().copy(), m.flat_grads().detach().cpu().numpy().copy()

    ref = one_step(None)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29573")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        queues = os.environ.get("GPU_MAX_HW_QUEUES")
        for mode in (False, True):
            if mode:
                # eight hardware queues (the package's default) + the overlapped exchange is the 2.5x-slower combination: refused
                os.environ["GPU_MAX_HW_QUEUES"] = "8"
                with pytest.raises(RuntimeError, match="GPU_MAX_HW_QUEUES"):
                    one_step(mode)
                os.environ["GPU_MAX_HW_QUEUES"] = "6"     # (the check reads the variable; HIP read it when it started: results do not depend on it)
            got = one_step(mode)
            for a, b in zip(got, ref):
                np.testing.assert_array_equal(a, b)
    finally:
        if queues is None:
            os.environ.pop("GPU_MAX_HW_QUEUES", None)
        else:
            os.environ["GPU_MAX_HW_QUEUES"] = queues
        dist.destroy_process_group()
    # the entry points refuse to run without a communicator, loudly
    m = make_model(H, L, gen, "f32", p)
    m._context(B)
    offs = (C.c_int64 * 1)(0); sizes = (C.c_int64 * 1)(16)
    assert _lib.lib().vae_allreduce_grads(m._ctx.handle, m.flat_grads().data_ptr(), 1, offs, sizes, 1, torch.cuda.current_stream().cuda_stream) != 0
    assert b"no communicator" in _lib.lib().vae_last_error()


@pytest.mark.parametrize("dtype,B", [("bf16", 3), ("f16", 5), ("bf16", 40)])   # (40 images x 8 bands = 320 units: more than one per workgroup)
def test_streaming_first_conv_matches_tiled_kernel(dtype, B):
    """encoder.1's forward on 128x128 images runs as a row-streaming kernel (dnfirst_stream.cuh: LDS-DMA ring in even / odd pixel planes,
    transposed MFMAs with the weights in registers).  Same products and rounding points as the tiled kernel (down2_kernel); the f32
    accumulation runs in two interleaved chains and adds the bias last, so a few stored values per 10^5 round the other way by one
    storage ulp and everything downstream follows at that level."""
    from torch_vae_amd import _lib
    H, L = 128, 16
    p = perturbed_params(L, H, 22, True)
    x = torch.from_numpy(vo.synth_pianoroll(B, H, 9)).cuda()
    eps = torch.from_numpy(vo.counter_normal(B * L, 9, 5).reshape(B, L)).float().cuda()
    res = []
    for stream in (0, 1):
        m = make_model(H, L, True, dtype, p, kld_weight=1.0)
        _lib.check(_lib.lib().vae_set_option(m._context(B).handle, b"use_dnf_stream", stream), "set")
        out3, xhat = m.fused_forward_backward(x, eps=eps)
        n = B * 64 * (H // 4) ** 2
        y1 = torch.empty(n, device="cuda")
        _lib.check(_lib.lib().vae_debug_tensor(m._ctx.handle, 1, y1.data_ptr(), n, torch.cuda.current_stream().cuda_stream), "dbg")
        torch.cuda.synchronize()
        res.append((out3.cpu().numpy(), xhat.cpu().numpy(), y1.cpu().numpy(), m.flat_grads().cpu().numpy(), m._bnflat.cpu().numpy()))
    (o0, x0, y0, g0, b0), (o1, x1, y1, g1, b1) = res
    frac = float((y0 != y1).mean())
    ulp = {"bf16": 2.0 ** -7, "f16": 2.0 ** -10}[dtype]
    excess = float((np.abs(y0 - y1) - 1.01 * ulp * np.abs(y0)).max() / np.abs(y0).max())
    report(test="streaming_first_conv", dtype=dtype, batch=B, differing_fraction=frac, excess_over_one_ulp=excess, grads=rel_l2(g1, g0))
    assert frac < 2e-3 and excess <= 1e-6, (frac, excess)
    np.testing.assert_allclose(o1, o0, rtol=5e-5)
    assert float(np.abs(x1 - x0).max()) < 5e-3      # (seven layers downstream of the perturbed roundings: measured 1.3e-3 in f16)
    assert rel_l2(b1, b0) < 1e-5
    assert rel_l2(g1, g0) < {"bf16": 1e-2, "f16": 5e-3}[dtype], rel_l2(g1, g0)


@pytest.mark.parametrize("dtype,H,L,B,gen", [("bf16", 128, 16, 37, True), ("f16", 64, 128, 70, True), ("bf16", 32, 10, 5, False), ("f16", 32, 16, 256, False)])
def test_wide_fc_input_gradient_matches_one_channel_per_thread_kernel(dtype, H, L, B, gen):
    """fc_mu | fc_var input gradient with 16-byte accesses (edge_kernels.cuh: fc_dgrad8_kernel, 8 channels x 4 rows per thread) against
    the one-channel-per-thread kernel: the stored dz of encoder.3 is bit-identical (same fma chain per element); its BatchNorm
    statistics are summed in another order, so the encoder gradients downstream agree to rounding."""
    from torch_vae_amd import _lib
    p = perturbed_params(L, H, 31, gen)
    x = torch.from_numpy(vo.synth_pianoroll(B, H, 14)).cuda()
    eps = torch.from_numpy(vo.counter_normal(B * L, 14, 5).reshape(B, L)).float().cuda()
    res = []
    for wide in (0, 1, 3):     # one channel per thread; 256-thread workgroups over 32 rows; 512-thread workgroups over 64 rows
        m = make_model(H, L, gen, dtype, p, kld_weight=1.0)
        _lib.check(_lib.lib().vae_set_option(m._context(B).handle, b"use_fc_dgrad8", wide), "set")
        out3, xhat = m.fused_forward_backward(x, eps=eps)
        s = (H // 16) if gen else 2
        n = B * 256 * s * s
        dz3 = torch.empty(n, device="cuda")
        _lib.check(_lib.lib().vae_debug_tensor(m._ctx.handle, 11, dz3.data_ptr(), n, torch.cuda.current_stream().cuda_stream), "dbg")
        torch.cuda.synchronize()
        res.append((out3.cpu().numpy(), dz3.cpu().numpy(), m.flat_grads().cpu().numpy()))
    (o0, d0, g0) = res[0]
    assert np.abs(d0).max() > 0
    for (o1, d1, g1) in res[1:]:
        assert np.array_equal(o0, o1)
        assert np.array_equal(d0, d1)
        report(test="wide_fc_dgrad", dtype=dtype, img=H, latent=L, batch=B, grads=rel_l2(g1, g0))
        assert rel_l2(g1, g0) < {"bf16": 5e-3, "f16": 2e-3}[dtype], rel_l2(g1, g0)


@pytest.mark.parametrize("dtype", ["bf16", "f16"])
def test_streaming_forward_kernels_in_eval_mode(dtype):
    """The streaming forward kernels (encoder.1, final_layer.0, decoder.2) with BatchNorm coefficients from RUNNING statistics (eval-mode
    forward, evaluation.evaluate's path) instead of the fused batch-statistics finalisation: same output as the tiled kernels."""
    from torch_vae_amd import _lib
    H, L, B = 128, 16, 3
    p = perturbed_params(L, H, 23, True)
    x = torch.from_numpy(vo.synth_pianoroll(B, H, 10)).cuda()
    eps = torch.from_numpy(vo.counter_normal(B * L, 10, 5).reshape(B, L)).float().cuda()
    outs = []
    for stream in (0, 1):
        m = make_model(H, L, True, dtype, p, kld_weight=1.0)
        h = m._context(B).handle
        _lib.check(_lib.lib().vae_set_option(h, b"use_upf_stream", 3 * stream), "set")
        _lib.check(_lib.lib().vae_set_option(h, b"use_dnf_stream", stream), "set")
        m.fused_forward_backward(x, eps=eps)          # one training step's forward: running statistics move away from (0, 1)
        m.eval()
        with torch.no_grad():
            m.set_next_eps(eps)
            out = m(x)
        outs.append((out["output"].float().cpu().numpy(), out["encoded"]["mu"].float().cpu().numpy()))
    (x0, mu0), (x1, mu1) = outs
    np.testing.assert_allclose(mu1, mu0, rtol=2e-2, atol=2e-3)
    assert float(np.abs(x1 - x0).max()) < 5e-3, float(np.abs(x1 - x0).max())
    assert rel_l2(x1, x0) < 1e-3


@pytest.mark.parametrize("bit,which", [(1, 7), (2, 6)])      # final_layer.0 (default on), decoder.2 (same template, 64 input channels; off by default: no gain)
@pytest.mark.parametrize("dtype,B", [("bf16", 3), ("f16", 5), ("bf16", 40)])   # (40 images x 8 bands = 320 units: more than one per workgroup)
def test_streaming_final_convt_matches_tiled_kernel(dtype, B, bit, which):
    """final_layer.0's forward on 128x128 images runs as a row-streaming kernel (upfinal_stream.cuh: LDS-DMA ring, transposed MFMAs,
    bands of rows per workgroup); the same template also covers decoder.2 (64 input channels, 32-pixel rows; off by default: no gain).  Same products and the same rounding points as the tiled kernel (up2_kernel); the f32 accumulation
    visits the taps in another order and adds the bias last, so a few stored values per 10^5 round the other way (measured
    0.003-0.04 %, always by one storage ulp) and everything downstream follows at that level."""
    from torch_vae_amd import _lib
    H, L = 128, 16
    p = perturbed_params(L, H, 21, True)
    x = torch.from_numpy(vo.synth_pianoroll(B, H, 8)).cuda()
    eps = torch.from_numpy(vo.counter_normal(B * L, 8, 5).reshape(B, L)).float().cuda()
    res = []
    for stream in (0, 1):
        m = make_model(H, L, True, dtype, p, kld_weight=1.0)
        _lib.check(_lib.lib().vae_set_option(m._context(B).handle, b"use_upf_stream", bit * stream), "set")
        out3, xhat = m.fused_forward_backward(x, eps=eps)
        n = B * 32 * (H if which == 7 else H // 2) ** 2
        y7 = torch.empty(n, device="cuda")
        _lib.check(_lib.lib().vae_debug_tensor(m._ctx.handle, which, y7.data_ptr(), n, torch.cuda.current_stream().cuda_stream), "dbg")
        torch.cuda.synchronize()
        res.append((out3.cpu().numpy(), xhat.cpu().numpy(), y7.cpu().numpy(), m.flat_grads().cpu().numpy(), m._bnflat.cpu().numpy()))
    (o0, x0, y0, g0, b0), (o1, x1, y1, g1, b1) = res
    frac = float((y0 != y1).mean())
    ulp = {"bf16": 2.0 ** -7, "f16": 2.0 ** -10}[dtype]
    # one storage rounding step where the value is of normal size; where the taps cancel (|y| tiny against its terms) the two
    # summation orders differ by f32 rounding of the terms instead: 1e-6 of the tensor's largest magnitude bounds that
    excess = float((np.abs(y0 - y1) - 1.01 * ulp * np.abs(y0)).max() / np.abs(y0).max())
    report(test="streaming_final_convt", layer=which, dtype=dtype, batch=B, differing_fraction=frac, excess_over_one_ulp=excess, grads=rel_l2(g1, g0))
    assert frac < 2e-3 and excess <= 1e-6, (frac, excess)   # (measured 1e-8 .. 2.5e-8)
    np.testing.assert_allclose(o1, o0, rtol=2e-5)
    assert float(np.abs(x1 - x0).max()) < {"bf16": 5e-3, "f16": 1e-3}[dtype]
    assert rel_l2(b1, b0) < 1e-6
    assert rel_l2(g1, g0) < {"bf16": 5e-3, "f16": 1e-3}[dtype], rel_l2(g1, g0)


@pytest.mark.parametrize("cfg", [(64, 16, 5, "bf16"), (128, 16, 3, "f16"), (32, 16, 9, "bf16"), (64, 64, 2, "f16")])
def test_fused_dgrad_wgrad_matches_separate_kernels(cfg):
    """conv_fused.cuh (one pass over (dz, y) for the input AND the weight gradient of final_layer.0 / decoder.2) against the
    separate input-gradient and weight-gradient kernels.  Same operands and the same MFMA accumulation order for the input
    gradient, so the dz it writes is bit-identical; the weight gradient and the BatchNorm statistics are summed in a different
    order (rounding level); everything downstream sees those last-bit differences through 16-bit roundings."""
    from torch_vae_amd import _lib
    H, L, B, dtype = cfg
    p = perturbed_params(L, H, 14, True)
    x = torch.from_numpy(vo.synth_pianoroll(B, H, 19)).cuda()
    eps = torch.from_numpy(vo.counter_normal(B * L, 19, 5).reshape(B, L)).float().cuda()
    res = []
    # separate kernels; all fused kernels with the stored dz; fused with final_layer.0's dz recomputed from dlogit (never stored);
    # decoder-side fused kernels only (the encoder.1 kernel's reference: identical inputs reach encoder.1 in variants 1 and 3)
    for use, recomp in ((0, 0), (1, 0), (1, 1), (2, 0)):
        model = make_model(H, L, True, dtype, p)
        model._context(B)
        assert _lib.lib().vae_set_option(model._ctx.handle, b"use_fused_wgrad", use) == 0
        assert _lib.lib().vae_set_option(model._ctx.handle, b"use_recomp_dz", recomp) == 0
        # (bit-identity across these variants needs the same BatchNorm-backward statistics in all of them: the row-streaming
        #  output-conv kernel sums them in another order than the tiled kernels the recomputing variant runs)
        assert _lib.lib().vae_set_option(model._ctx.handle, b"use_convout_stream", 0) == 0
        out3, _ = model.fused_forward_backward(x, eps=eps)
        n = B * 32 * (H // 2) ** 2
        dz6 = torch.empty(n, device="cuda")
        _lib.check(_lib.lib().vae_debug_tensor(model._ctx.handle, 8 + 6, dz6.data_ptr(), n, torch.cuda.current_stream().cuda_stream), "dbg")
        dz0 = torch.empty(n, device="cuda")                                   # dz of encoder.0's output, written by the encoder.1 kernel
        _lib.check(_lib.lib().vae_debug_tensor(model._ctx.handle, 8 + 0, dz0.data_ptr(), n, torch.cuda.current_stream().cuda_stream), "dbg")
        res.append((out3.cpu().numpy(), dz6.cpu().numpy(), flat_grad_dict(model), dz0.cpu().numpy()))
    for k in (1, 2):
        np.testing.assert_array_equal(res[0][0], res[k][0])
        np.testing.assert_array_equal(res[0][1], res[k][1])                   # dz of decoder.2's output, written by the layer-7 kernel
        g0, g1 = res[0][2], res[k][2]
        for n in ("final_layer.0.weight", "final_layer.1.weight", "final_layer.1.bias", "final_layer.3.weight", "decoder.2.1.weight", "decoder.2.1.bias"):
            assert rel_l2(g1[n], g0[n]) < 2e-5, (k, n, rel_l2(g1[n], g0[n]))
        for n in g0:
            if n not in PRE_BN_BIAS:
                assert rel_l2(g1[n], g0[n]) < (2e-2 if dtype == "bf16" else 5e-3), (k, n, rel_l2(g1[n], g0[n]))
        # encoder.1's fused kernel sees inputs that differ in the last bits (statistics summed in another order upstream), so
        # its dz is compared at rounding level, not bit for bit
        assert rel_l2(res[k][3], res[0][3]) < (2e-2 if dtype == "bf16" else 5e-3)
    np.testing.assert_array_equal(res[1][3], res[3][3])                       # encoder.1 fused vs separate on identical inputs: bit-identical dz
    assert rel_l2(res[1][2]["encoder.1.0.weight"], res[3][2]["encoder.1.0.weight"]) < 2e-5
    assert rel_l2(res[1][2]["encoder.0.1.weight"], res[3][2]["encoder.0.1.weight"]) < 2e-5


@pytest.mark.parametrize("cfg", [(64, 16, 5, "bf16"), (128, 16, 3, "f16"), (32, 16, 33, "bf16")])
def test_materialised_operands_weight_gradients_match(cfg):
    """Deep layers (encoder.2/3, decoder.0/1): the weight-gradient kernel reading the MATERIALISED operands (LeakyReLU(BN(y)) written by
    the forward kernel that stages it, the BatchNorm-backward gradient written by the input-gradient kernel) against the same
    kernel transforming its operands itself: identical operand values, identical tiling - nothing in the step may change."""
    from torch_vae_amd import _lib
    H, L, B, dtype = cfg
    p = perturbed_params(L, H, 31, True)
    x = torch.from_numpy(vo.synth_pianoroll(B, H, 37)).cuda()
    eps = torch.from_numpy(vo.counter_normal(B * L, 37, 5).reshape(B, L)).float().cuda()
    res = []
    for raw in (0, 1):
        model = make_model(H, L, True, dtype, p)
        model._context(B)
        assert _lib.lib().vae_set_option(model._ctx.handle, b"use_raw_wgrad", raw) == 0
        # (operands are materialised by the pipelined kernels' staging: the workgroup-specialised deep-layer kernels, which sum
        #  their BatchNorm statistics in another order, are switched off on both sides of this bit-for-bit comparison)
        assert _lib.lib().vae_set_option(model._ctx.handle, b"use_deep", 0) == 0
        out3, _ = model.fused_forward_backward(x, eps=eps)
        res.append((out3.cpu().numpy(), flat_grad_dict(model)))
    np.testing.assert_array_equal(res[0][0], res[1][0])
    for n in res[0][1]:
        if n in ("encoder.2.0.weight", "encoder.3.0.weight", "decoder.0.0.weight", "decoder.1.0.weight"):
            assert rel_l2(res[1][1][n], res[0][1][n]) < 1e-6, (n, rel_l2(res[1][1][n], res[0][1][n]))
        else:
            np.testing.assert_array_equal(res[1][1][n], res[0][1][n], err_msg=n)


@pytest.mark.parametrize("dtype", ["bf16", "f16"])
def test_deferred_output_conv_matches_separate_kernels(dtype):
    """The fused step's forward runs with train = 2: output conv + sigmoid + BCE are left to the backward, where ONE kernel
    does that layer's forward and backward in a single pass over y7 (convout_step_mfma_kernel).  Same arithmetic element
    for element as convout_fwd_mfma_kernel + convout_bwd_mfma_kernel: xhat, the ELBO scalars, dz of final_layer's
    BatchNorm and every gradient must be bit-identical; BatchNorm running statistics too."""
    from torch_vae_amd import _lib
    H, L, B = 64, 16, 6
    p = perturbed_params(L, H, 11, True)
    x = torch.from_numpy(vo.synth_pianoroll(B, H, 5)).cuda()
    eps = torch.from_numpy(vo.counter_normal(B * L, 5, 5).reshape(B, L)).float().cuda()
    res = []
    for fused in (0, 1):
        m = make_model(H, L, True, dtype, p, kld_weight=2.0)
        _lib.check(_lib.lib().vae_set_option(m._context(B).handle, b"use_fused_convout", fused), "set")
        out3, xhat = m.fused_forward_backward(x, eps=eps)
        dz7 = torch.empty(B * 32 * H * H, device="cuda")
        _lib.check(_lib.lib().vae_debug_tensor(m._ctx.handle, 15, dz7.data_ptr(), dz7.numel(), torch.cuda.current_stream().cuda_stream), "dbg")
        res.append((out3.clone(), xhat.clone(), dz7, m.flat_grads().clone(), m._bnflat.clone()))
        if fused:     # the deferred forward can be differentiated once, and only through the standard ELBO
            L_, st = _lib.lib(), torch.cuda.current_stream().cuda_stream
            assert L_.vae_backward(m._ctx.handle, x.data_ptr(), m._flat.data_ptr(), m._gflat.data_ptr(), 0, 0, 0, 0, 0, 0, 2.0, 1, st) != 0
            # a pending train = 2 forward refuses vae_loss (the BCE does not exist yet) and a backward with a loss scale
            xh, mu_, lv_, z_ = (torch.empty_like(x), *(torch.empty(B, L, device="cuda") for _ in range(3)))
            assert L_.vae_forward(m._ctx.handle, x.data_ptr(), B, m._flat.data_ptr(), m._bnflat.data_ptr(), m._nbt.data_ptr(),
                                  eps.data_ptr(), 0, 2, xh.data_ptr(), mu_.data_ptr(), lv_.data_ptr(), z_.data_ptr(), st) == 0
            o3 = torch.empty(3, device="cuda"); two = torch.full((1,), 2.0, device="cuda")
            assert L_.vae_loss(m._ctx.handle, 2.0, o3.data_ptr(), st) != 0
            assert L_.vae_backward(m._ctx.handle, x.data_ptr(), m._flat.data_ptr(), m._gflat.data_ptr(), 0, two.data_ptr(), 0, 0, 0, 0, 2.0, 1, st) != 0
            assert L_.vae_loss_deferred(m._ctx.handle, 2.0, o3.data_ptr(), st) == 0
            assert L_.vae_backward(m._ctx.handle, x.data_ptr(), m._flat.data_ptr(), m._gflat.data_ptr(), 0, 0, 0, 0, 0, 0, 2.0, 1, st) == 0
            torch.cuda.synchronize()
            assert torch.equal(o3, out3) and torch.equal(xh, xhat)
    (o0, x0, d0, g0, b0), (o1, x1, d1, g1, b1) = res
    assert torch.equal(x0, x1) and torch.equal(o0, o1)
    assert torch.equal(d0, d1)
    assert torch.equal(b0, b1)
    from torch_vae_amd._lib import PARAM_NAMES
    offs, sizes = m._offs, m._sizes
    bad = {}
    for n, o, sz in zip(PARAM_NAMES, offs, sizes):
        if not torch.equal(g0[o:o + sz], g1[o:o + sz]):
            bad[n] = float((g0[o:o + sz] - g1[o:o + sz]).norm() / g0[o:o + sz].norm())
    report(test="deferred_output_conv", dtype=dtype, differing=bad)
    assert not bad, bad


@pytest.mark.parametrize("dtype", ["bf16", "f16"])
@pytest.mark.parametrize("B,bands", [(3, 0), (2, 1), (5, 4), (2, 16), (20, 16)])   # (20 x 16 bands = 320 units: workgroups walk more than one)
def test_streaming_output_conv_matches_tiled_kernel(dtype, B, bands):
    """128-pixel-wide images take the row-streaming form of the fused output-conv kernel (convout_stream.cuh: LDS rings filled by
    LDS-DMA, transposed MFMAs, whole rows or bands of rows per workgroup).  Same arithmetic element for element as the tiled kernel
    (convout_step_mfma_kernel): xhat, dz of final_layer's BatchNorm and the ELBO's reconstruction term must be bit-identical for
    every band split; sums taken in another order (the BatchNorm-backward statistics, the conv's own weight gradient, the BCE
    total) agree to f32 rounding, and with them every gradient upstream."""
    from torch_vae_amd import _lib
    H, L = 128, 16
    p = perturbed_params(L, H, 13, True)
    x = torch.from_numpy(vo.synth_pianoroll(B, H, 6)).cuda()
    eps = torch.from_numpy(vo.counter_normal(B * L, 6, 5).reshape(B, L)).float().cuda()
    res = []
    for stream in (0, 1):
        m = make_model(H, L, True, dtype, p, kld_weight=1.5)
        h = m._context(B).handle
        _lib.check(_lib.lib().vae_set_option(h, b"use_convout_stream", stream), "set")
        _lib.check(_lib.lib().vae_set_option(h, b"knob_convout_bands", bands), "set")
        out3, xhat = m.fused_forward_backward(x, eps=eps)
        dz7 = torch.empty(B * 32 * H * H, device="cuda")
        _lib.check(_lib.lib().vae_debug_tensor(m._ctx.handle, 15, dz7.data_ptr(), dz7.numel(), torch.cuda.current_stream().cuda_stream), "dbg")
        torch.cuda.synchronize()
        res.append((out3.clone(), xhat.clone(), dz7, m.flat_grads().clone(), m._bnflat.clone()))
    (o0, x0, d0, g0, b0), (o1, x1, d1, g1, b1) = res
    assert torch.equal(x0, x1)
    assert torch.equal(d0, d1)
    assert torch.equal(b0, b1)
    np.testing.assert_allclose(o1.cpu().numpy(), o0.cpu().numpy(), rtol=2e-6)
    from torch_vae_amd._lib import PARAM_NAMES
    worst = {}
    for n, o, sz in zip(PARAM_NAMES, m._offs, m._sizes):
        worst[n] = rel_l2(g1[o:o + sz].cpu().numpy(), g0[o:o + sz].cpu().numpy())
    report(test="streaming_output_conv", dtype=dtype, batch=B, bands=bands, worst=max(worst.values()))
    # (an f32 rounding step in a BatchNorm-backward statistic moves 16-bit rounding decisions of every gradient upstream of it, and
    #  at these tiny batches a handful of flipped roundings is visible in the small tensors - measured worst 8e-3 (bf16,
    #  encoder.0.1.bias): the bound is a few rounding steps of the storage type, not of f32)
    tol = {"bf16": 3e-2, "f16": 5e-3}[dtype]
    assert max(worst.values()) < tol, {k: v for k, v in worst.items() if v >= tol}
    assert max(worst[n] for n in ("final_layer.3.weight", "final_layer.3.bias", "final_layer.1.weight", "final_layer.1.bias")) < 1e-5, worst


def _layer_local_gaps(dtype, H, L, B, gen, seed=41, opts=None, kld_weight=1.0, exact_convout=False):
    """Every 16-bit kernel of the step against the storage-emulating oracle ON THE KERNEL'S OWN INPUTS: each stored tensor (y_l, dz_l,
    decoder_input's output and gradient) and each parameter gradient is recomputed on the CPU from the tensors the GPU actually
    stored one layer earlier, so a gap is that one kernel's, not the chain's.  (The end-to-end emulation of
    test_every_tensor_against_oracle cannot be tighter than the chain allows: 1-ulp differences in stored activations flip
    LeakyReLU slopes a few layers later - DESIGN.md section 4.)  Returns {tensor name: relative L2 gap}."""
    from torch_vae_amd import _lib
    p = perturbed_params(L, H, seed, gen)
    x = vo.synth_pianoroll(B, H, 21).astype(np.float64)
    eps = vo.counter_normal(B * L, 21, 5).reshape(B, L).astype(np.float64)
    m = make_model(H, L, gen, dtype, p, kld_weight=kld_weight)
    for k, v in (opts or {}).items():
        _lib.check(_lib.lib().vae_set_option(m._context(B).handle, k.encode(), v), "set " + k)
    m.fused_forward_backward(torch.from_numpy(x).float().cuda(), eps=torch.from_numpy(eps).float().cuda())
    torch.cuda.synchronize()
    grads = flat_grad_dict(m)
    last = {k: m._last[k].double().cpu().numpy() for k in ("xhat", "mu", "lv", "z")}
    C = [32, 64, 128, 256, 128, 64, 32, 32]
    s = H // 16 if gen else 2
    HW = [H // 2, H // 4, H // 8, s, 2 * s, 4 * s, 8 * s, 16 * s]
    gs = vo.f16_grad_scale(B, H) if dtype == "f16" else 1.0
    st = torch.cuda.current_stream().cuda_stream

    def fetch(which, shape, scale=1.0):
        n = int(np.prod(shape))
        t = torch.empty(n, device="cuda")
        _lib.check(_lib.lib().vae_debug_tensor(m._ctx.handle, which, t.data_ptr(), n, st), "dbg")
        torch.cuda.synchronize()
        return t.cpu().numpy().reshape(shape).astype(np.float64) / scale

    Y = [fetch(i, (B, C[i], HW[i], HW[i])) for i in range(8)]
    DZ = [fetch(8 + i, (B, C[i], HW[i], HW[i]), gs) for i in range(8)]
    d0 = fetch(16, (B, 256, s, s))
    dd0 = fetch(17, (B, 256, s, s), gs)
    names = ["encoder.0", "encoder.1", "encoder.2", "encoder.3", "decoder.0", "decoder.1", "decoder.2", "final_layer"]
    store = None if dtype == "f32" else dtype            # (f32 mode: nothing is rounded to 16 bits; the pre-activation is still the f32 fma)
    rs = lambda v: vo.round_storage(v, store)            # noqa: E731
    rg = lambda v: vo.round_storage(v, store, gs)        # noqa: E731
    P = lambda k: p[k].astype(np.float64)                # noqa: E731
    Z, CA = [], []
    for i in range(8):      # pre-activations and BatchNorm caches of the GPU's stored y_l
        z, cache = vo.bn_train_fwd_stored(Y[i], P(names[i] + ".1.weight"), P(names[i] + ".1.bias"), dtype)
        Z.append(z); CA.append(cache)
    A = [rs(vo.lrelu(z)) for z in Z]                      # staged operands LeakyReLU(BN(y_l)) as the next kernel rounds them
    gaps = {}
    G = lambda k: grads[k].reshape(p[k].shape).astype(np.float64)   # noqa: E731
    # ---- forward, layer by layer on the GPU's own inputs
    gaps["y0"] = rel_l2(rs(vo.conv_fwd(x, P("encoder.0.0.weight"), P("encoder.0.0.bias"), 2)), Y[0])
    for i in (1, 2, 3):
        gaps[f"y{i}"] = rel_l2(rs(vo.conv_fwd(A[i - 1], rs(P(names[i] + ".0.weight")), P(names[i] + ".0.bias"), 2)), Y[i])
    pre = A[3].reshape(B, -1)
    gaps["mu"] = rel_l2(pre @ rs(P("fc_mu.weight")).T + P("fc_mu.bias"), last["mu"])
    gaps["log_var"] = rel_l2(pre @ rs(P("fc_var.weight")).T + P("fc_var.bias"), last["lv"])
    gaps["z"] = rel_l2(eps * np.exp(0.5 * last["lv"]) + last["mu"], last["z"])
    gaps["d0"] = rel_l2(rs(last["z"] @ P("decoder_input.weight").T + P("decoder_input.bias")).reshape(d0.shape), d0)
    ins = {4: d0, 5: A[4], 6: A[5], 7: A[6]}
    for i in (4, 5, 6, 7):
        gaps[f"y{i}"] = rel_l2(rs(vo.convT_fwd(ins[i], rs(P(names[i] + ".0.weight")), P(names[i] + ".0.bias"))), Y[i])
    # the output conv: the MFMA kernels multiply 16-bit operands (staged a7, packed weights, dlogit); the VALU kernels kept as the
    # fallback (use_mfma_convout = 0; also the f32 mode's path) stage a7 in f32 and read the f32 weights and dlogit (exact_convout)
    a7 = vo.lrelu(Z[7]) if exact_convout else A[7]
    wo = P("final_layer.3.weight") if exact_convout else rs(P("final_layer.3.weight"))
    gaps["xhat"] = rel_l2(vo.sigmoid(vo.conv_fwd(a7, wo, P("final_layer.3.bias"), 1)), last["xhat"])
    # ---- backward: the output conv from the GPU's xhat, then every block from the GPU's stored (dz_l, y_l)
    xh = last["xhat"]
    dlogit = (xh - x) / np.maximum(xh * (1 - xh), 1e-12) / xh.size * xh * (1 - xh)
    dl_op = dlogit if exact_convout else rg(dlogit)
    dw, _ = vo.conv_wgrad(a7, dl_op, 1)
    gaps["final_layer.3.weight"] = rel_l2(dw, G("final_layer.3.weight"))
    gaps["final_layer.3.bias"] = rel_l2(dlogit.sum(axis=(0, 2, 3)), G("final_layer.3.bias"))
    gaps["dz7"] = rel_l2(rg(vo.lrelu_bwd(Z[7], vo.conv_dgrad(dl_op, wo, 1, (H, H)))), DZ[7])
    for i in (7, 6, 5, 4, 3, 2, 1, 0):
        n = names[i]
        dy, dgam, dbet = vo.bn_train_bwd(DZ[i], P(n + ".1.weight"), CA[i])
        gaps[n + ".1.weight"] = rel_l2(dgam, G(n + ".1.weight"))
        gaps[n + ".1.bias"] = rel_l2(dbet, G(n + ".1.bias"))
        if i == 0:          # conv1_wgrad: f32 weights, unrounded gradient operand
            dw, db = vo.conv_wgrad(x, dy, 2)
            gaps[n + ".0.weight"] = rel_l2(dw, G(n + ".0.weight"))
            continue
        dyr = rg(dy)
        w = rs(P(n + ".0.weight"))
        if i >= 4:
            dx, dw, db = vo.convT_bwd(ins[i], w, dyr)
        else:
            dw, db = vo.conv_wgrad(A[i - 1], dyr, 2)
            dx = vo.conv_dgrad(dyr, w, 2, A[i - 1].shape[2:])
        gaps[n + ".0.weight"] = rel_l2(dw, G(n + ".0.weight"))
        if i == 4:
            gaps["dd0"] = rel_l2(rg(dx), dd0)
            # latent block on the GPU's dd0, mu, log_var, z
            f = dd0.reshape(B, -1)
            gaps["decoder_input.weight"] = rel_l2(f.T @ last["z"], G("decoder_input.weight"))
            gaps["decoder_input.bias"] = rel_l2(f.sum(axis=0), G("decoder_input.bias"))
            dzl = f @ rs(P("decoder_input.weight"))
            mu, lv = last["mu"], last["lv"]
            dmu = dzl + kld_weight * mu / B
            dlv = dzl * eps * np.exp(0.5 * lv) * 0.5 + kld_weight * 0.5 * (np.exp(lv) - 1) / B
            act3 = vo.lrelu(Z[3]).reshape(B, -1)          # (unrounded: the fc weight gradient reads it in f32)
            gaps["fc_mu.weight"] = rel_l2(dmu.T @ act3, G("fc_mu.weight"))
            gaps["fc_var.weight"] = rel_l2(dlv.T @ act3, G("fc_var.weight"))
            gaps["fc_mu.bias"] = rel_l2(dmu.sum(axis=0), G("fc_mu.bias"))
            gaps["fc_var.bias"] = rel_l2(dlv.sum(axis=0), G("fc_var.bias"))
            dpre = dmu @ rs(P("fc_mu.weight")) + dlv @ rs(P("fc_var.weight"))
            gaps["dz3"] = rel_l2(rg(vo.lrelu_bwd(Z[3], dpre.reshape(Z[3].shape))), DZ[3])
        else:
            gaps[f"dz{i - 1}"] = rel_l2(rg(vo.lrelu_bwd(Z[i - 1], dx)), DZ[i - 1])
    return gaps


@pytest.mark.parametrize("dtype,H,L,B,gen", [("f32", 32, 16, 6, False), ("f32", 64, 16, 5, True), ("f32", 128, 16, 3, True),
                                             ("bf16", 32, 16, 6, False), ("f16", 32, 16, 6, False), ("bf16", 32, 10, 7, False),
                                             ("bf16", 64, 16, 5, True), ("f16", 64, 128, 4, True), ("bf16", 128, 16, 3, True),
                                             ("f16", 128, 16, 3, True), ("bf16", 128, 16, 9, True), ("f16", 128, 64, 8, True),
                                             # BASELINE configs[1] and configs[4] (per GPU) at full size; configs[2]'s model at batch 64
                                             ("bf16", 128, 16, 256, True), ("f16", 128, 128, 512 if FULL_TESTS else 64, True),
                                             ("bf16", 256, 64, 64 if FULL_TESTS else 16, True)])
def test_every_kernel_against_oracle_on_its_own_inputs(dtype, H, L, B, gen):
    """Layer-local parity of the 16-bit modes (see _layer_local_gaps): 54 tensors per case - every stored activation and gradient,
    the latent block, xhat and every parameter gradient - each within 5e-4 (relative L2) of the storage-emulating oracle evaluated
    on the kernel's own inputs.  Measured on MI355X: worst 1.0e-4 (a stored dz; 2.3e-4 on the 256x256 model), most tensors 1e-7 .. 3e-5, many bit-identical."""
    gaps = _layer_local_gaps(dtype, H, L, B, gen)
    worst = max(gaps, key=gaps.get)
    report(test="layer_local", dtype=dtype, img=H, latent=L, batch=B, worst=worst, worst_gap=gaps[worst], gaps=gaps)
    assert len(gaps) == 54
    gate = 1e-5 if dtype == "f32" else 5e-4       # f32 kernel mode: f32 against f64 accumulation only (measured: worst 6.2e-7)
    bad = {k: v for k, v in gaps.items() if not v < gate}
    assert not bad, bad


# Every alternative kernel the library can be switched to (tiled instead of streaming, separate instead of fused, the older
# generations kept as fallbacks for other shapes, diagnostic layouts), each through the same layer-local check as the defaults.
KERNEL_VARIANTS = [
    {"use_convout_stream": 0, "use_upf_stream": 0, "use_dnf_stream": 0},     # tiled forms of the three streaming kernels
    {"use_upf_stream": 3},                                                   # both transposed-conv forwards streaming
    {"use_fused_wgrad": 0},                                                  # separate input / weight gradient kernels on layers 7, 6, 1
    {"use_fused_wgrad": 1}, {"use_fused_wgrad": 2},
    {"use_wgrad_split": 0, "use_deep": 0},                                   # 8-wave weight gradients, down2 / up2 everywhere
    {"use_deep": 3},                                                         # dn3 + up3
    {"use_latent_mfma": 15},                                                 # the whole latent block on the exact-f32 MFMA
    {"use_latent_mfma": 0, "use_fc_dgrad8": 0},                              # ... and on the VALU kernels
    {"use_fc_dgrad8": 3},
    {"use_fused_convout": 0},                                                # output conv forward, BCE and backward as separate kernels
    {"use_pipelined": 0},                                                    # one-tile-per-workgroup conv kernels
    {"use_raw_wgrad": 1},                                                    # materialised weight-gradient operands
    {"use_side_stream": 0, "use_fused_bn": 0},                               # one stream, standalone BatchNorm finalisation launches
    {"use_tr16": 0},                                                         # weight gradients without the transposed LDS reads
    {"knob_wgrad_tile": 0}, {"knob_wgrad_wide": 0}, {"knob_wgrad_force_simple": 1},
    {"use_fused_convout": 0, "use_mfma_convout": 0},                         # the VALU output-conv kernels (forward and backward)
]


@pytest.mark.parametrize("vi", range(len(KERNEL_VARIANTS)), ids=["+".join(f"{k}={v}" for k, v in o.items()) for o in KERNEL_VARIANTS])
def test_every_kernel_variant_against_oracle_on_its_own_inputs(vi):
    """The layer-local check of test_every_kernel_against_oracle_on_its_own_inputs with the library switched to each alternative kernel:
    all 54 tensors within 5e-4 for every variant (measured on MI355X: worst 1.4e-4)."""
    opts = KERNEL_VARIANTS[vi]
    shapes = (("bf16", 128, 16, 9), ("f16", 64, 16, 5)) if (FULL_TESTS or vi % 3 == 0) else (("bf16", 128, 16, 9),)
    for dtype, H, L, B in shapes:
        gaps = _layer_local_gaps(dtype, H, L, B, True, seed=43, opts=opts, exact_convout=opts.get("use_mfma_convout", 1) == 0)
        worst = max(gaps, key=gaps.get)
        report(test="layer_local_variant", opts=opts, dtype=dtype, img=H, worst=worst, worst_gap=gaps[worst])
        bad = {k: v for k, v in gaps.items() if not v < 5e-4}
        assert not bad, (opts, dtype, bad)


@pytest.mark.parametrize("dtype,H,L,B,gen", [("f32", 64, 16, 5, True), ("bf16", 32, 16, 6, False), ("f16", 64, 16, 5, True),
                                             ("bf16", 128, 16, 9, True), ("f16", 128, 16, 3, True)])
def test_eval_mode_forward_kernels_against_oracle_on_their_own_inputs(dtype, H, L, B, gen):
    """The forward with BatchNorm in running-statistics mode (model.eval(): evaluation.evaluate's path, evaluation.py:12-113 of the
    reference) through the same layer-local check: every stored y_l, mu, log_var, z, decoder_input's output and x_hat recomputed on
    the CPU from what the GPU stored one layer earlier and the running statistics it holds."""
    from torch_vae_amd import _lib
    p = perturbed_params(L, H, 47, gen)
    x = vo.synth_pianoroll(B, H, 23).astype(np.float64)
    eps = vo.counter_normal(B * L, 23, 5).reshape(B, L).astype(np.float64)
    xt, et = torch.from_numpy(x).float().cuda(), torch.from_numpy(eps).float().cuda()
    m = make_model(H, L, gen, dtype, p, kld_weight=1.0)
    m.fused_forward_backward(xt, eps=et)            # a training-mode forward first: the running statistics move away from (0, 1)
    m.eval()
    with torch.no_grad():
        m.set_next_eps(et)
        out = m(xt)
    torch.cuda.synchronize()
    got = {"xhat": out["output"], "mu": out["encoded"]["mu"], "lv": out["encoded"]["log_var"], "z": out["latents"]}
    got = {k: v.double().cpu().numpy() for k, v in got.items()}
    C = [32, 64, 128, 256, 128, 64, 32, 32]
    s = H // 16 if gen else 2
    HW = [H // 2, H // 4, H // 8, s, 2 * s, 4 * s, 8 * s, 16 * s]
    st = torch.cuda.current_stream().cuda_stream

    def fetch(which, shape):
        n = int(np.prod(shape))
        t = torch.empty(n, device="cuda")
        _lib.check(_lib.lib().vae_debug_tensor(m._ctx.handle, which, t.data_ptr(), n, st), "dbg")
        torch.cuda.synchronize()
        return t.cpu().numpy().reshape(shape).astype(np.float64)

    Y = [fetch(i, (B, C[i], HW[i], HW[i])) for i in range(8)]
    d0 = fetch(16, (B, 256, s, s))
    names = ["encoder.0", "encoder.1", "encoder.2", "encoder.3", "decoder.0", "decoder.1", "decoder.2", "final_layer"]
    store = None if dtype == "f32" else dtype
    rs = lambda v: vo.round_storage(v, store)            # noqa: E731
    P = lambda k: p[k].astype(np.float64)                # noqa: E731
    bnflat = m._bnflat.double().cpu().numpy()
    A = []
    for i in range(8):
        o, c = m._bn_offs[i], m._bn_ch[i]
        rm, rv = bnflat[o:o + c], bnflat[o + c:o + 2 * c]
        assert np.abs(rm).max() > 0                      # (moved by the training step above)
        A.append(rs(vo.lrelu(vo.bn_eval_fwd_stored(Y[i], P(names[i] + ".1.weight"), P(names[i] + ".1.bias"), rm, rv, dtype))))
    gaps = {"y0": rel_l2(rs(vo.conv_fwd(x, P("encoder.0.0.weight"), P("encoder.0.0.bias"), 2)), Y[0])}
    for i in (1, 2, 3):
        gaps[f"y{i}"] = rel_l2(rs(vo.conv_fwd(A[i - 1], rs(P(names[i] + ".0.weight")), P(names[i] + ".0.bias"), 2)), Y[i])
    pre = A[3].reshape(B, -1)
    gaps["mu"] = rel_l2(pre @ rs(P("fc_mu.weight")).T + P("fc_mu.bias"), got["mu"])
    gaps["log_var"] = rel_l2(pre @ rs(P("fc_var.weight")).T + P("fc_var.bias"), got["lv"])
    gaps["z"] = rel_l2(eps * np.exp(0.5 * got["lv"]) + got["mu"], got["z"])
    gaps["d0"] = rel_l2(rs(got["z"] @ P("decoder_input.weight").T + P("decoder_input.bias")).reshape(d0.shape), d0)
    ins = {4: d0, 5: A[4], 6: A[5], 7: A[6]}
    for i in (4, 5, 6, 7):
        gaps[f"y{i}"] = rel_l2(rs(vo.convT_fwd(ins[i], rs(P(names[i] + ".0.weight")), P(names[i] + ".0.bias"))), Y[i])
    gaps["xhat"] = rel_l2(vo.sigmoid(vo.conv_fwd(A[7], rs(P("final_layer.3.weight")), P("final_layer.3.bias"), 1)), got["xhat"])
    worst = max(gaps, key=gaps.get)
    report(test="layer_local_eval", dtype=dtype, img=H, latent=L, batch=B, worst=worst, worst_gap=gaps[worst], gaps=gaps)
    gate = 1e-5 if dtype == "f32" else 5e-4          # measured on MI355X: 7.1e-7 (f32), <= 5.2e-5 (16-bit)
    bad = {k: v for k, v in gaps.items() if not v < gate}
    assert not bad, bad


@pytest.mark.parametrize("dtype,kw", [("f16", 16.0), ("bf16", 4.0), ("f32", 16.0)])
def test_every_kernel_against_oracle_on_its_own_inputs_beta_vae(dtype, kw):
    """BASELINE configs[4]'s beta-VAE weights (kld_weight 4 / 16, latent 128) through the layer-local check."""
    gaps = _layer_local_gaps(dtype, 128, 128, 6, True, seed=53, kld_weight=kw)
    worst = max(gaps, key=gaps.get)
    report(test="layer_local_beta", dtype=dtype, kld_weight=kw, worst=worst, worst_gap=gaps[worst])
    gate = 1e-5 if dtype == "f32" else 5e-4
    bad = {k: v for k, v in gaps.items() if not v < gate}
    assert not bad, bad


@pytest.mark.parametrize("dtype,wd", [("bf16", 0.0), ("f32", 0.01)])
def test_adamw_kernel_against_oracle_on_its_own_inputs(dtype, wd):
    """The fused AdamW + OneCycle update (train.py:228-238, 656-659) in the same spirit: at the third step of the one-call training
    step the new parameters and both moments are recomputed on the CPU (f64) from the GPU's own gradients, old parameters, old
    moments and the scheduler's (lr, beta1) of that step; the modules the reference never optimises must not move."""
    from argparse import Namespace
    from torch_vae_amd import _lib
    from torch_vae_amd.train import build_optimizer, fused_step
    H, L, B = 64, 16, 6
    p = perturbed_params(L, H, 61, True)
    m = make_model(H, L, True, dtype, p, kld_weight=1.0)
    cfg = Namespace(batch_size_per_gpu=B, world_size=1, lr_relative=0.01, weight_decay=wd, optimizer="AdamW", scheduler="OneCycle", epochs=1,
                    freeze_encoder=False)
    opt, sched = build_optimizer(cfg, m, steps_per_epoch=10)
    opt._bind()                                        # (the flat moment buffers are created on first use)
    worst = {}
    for step in range(1, 4):
        x = torch.from_numpy(vo.synth_pianoroll(B, H, 30 + step)).cuda()
        eps = torch.from_numpy(vo.counter_normal(B * L, 30 + step, 5).reshape(B, L)).float().cuda()
        p0 = m.flat_parameters().double().cpu().numpy().copy()
        m0, v0 = opt._m.double().cpu().numpy().copy(), opt._v.double().cpu().numpy().copy()
        hyper = [(g["lr"], g["betas"][0], g["betas"][1], g["eps"], g["weight_decay"]) for g in opt.param_groups]
        fused_step(m, opt, x, eps=eps)
        torch.cuda.synchronize()
        g = m.flat_grads().double().cpu().numpy()
        p1 = m.flat_parameters().double().cpu().numpy()
        m1, v1 = opt._m.double().cpu().numpy(), opt._v.double().cpu().numpy()
        touched = np.zeros(p0.size, dtype=bool)
        for (lr, b1, b2, e, wdg), (o, n) in zip(hyper, opt._ranges):
            sl = slice(o, o + n)
            touched[sl] = True
            # torch's scalars: every expression below in double from the Python floats, rounded ONCE to the float the element-wise
            # update multiplies with ((float)(1 - 0.999) = 0.001f) - the C ABI carries the hyper-parameters as doubles for that
            f = lambda v: float(np.float32(v))        # noqa: E731
            mm = m0[sl] * f(b1) + f(1 - b1) * g[sl]
            vv = v0[sl] * f(b2) + f(1 - b2) * g[sl] * g[sl]
            denom = np.sqrt(vv) * f(1 / np.sqrt(1 - b2 ** step)) + f(e)
            want = p0[sl] * f(1 - lr * wdg) - f(lr / (1 - b1 ** step)) * (mm / denom)
            for name, got_, want_ in (("param", p1[sl], want), ("exp_avg", m1[sl], mm), ("exp_avg_sq", v1[sl], vv)):
                worst[name] = max(worst.get(name, 0.0), rel_l2(got_, want_))
            # the update itself (p1 - p0 is ~1e-3 of p): compared on its own so that the parameter's magnitude cannot hide it
            worst["update"] = max(worst.get("update", 0.0), rel_l2(p1[sl] - p0[sl], want - p0[sl]))
        assert touched.sum() == sum(n for _, n in opt._ranges) and not touched.all()
        assert np.array_equal(p1[~touched], p0[~touched])          # fc_mu, fc_var, decoder_input, final_layer: never updated (train.py:210-225)
        sched.step()
    report(test="adamw_local", dtype=dtype, weight_decay=wd, **worst)
    assert worst["param"] < 1e-6 and worst["exp_avg"] < 1e-6 and worst["exp_avg_sq"] < 1e-6 and worst["update"] < 1e-4, worst
