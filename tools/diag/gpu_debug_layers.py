"""Per-layer diagnosis on the GPU box: compares every internal tensor of the HIP path with the
oracle and prints relative errors.  Not a pytest file; run as a script."""
import sys, os, json
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import vae_oracle as vo
from torch_vae_amd import _lib
from torch_vae_amd.models import VanillaVAE


def rel(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return float(np.sqrt(((a - b) ** 2).sum()) / max(np.sqrt((b ** 2).sum()), 1e-30))


def dbg(model, which, shape):
    out = torch.empty(int(np.prod(shape)), device="cuda")
    _lib.check(_lib.lib().vae_debug_tensor(model._ctx.handle, which, out.data_ptr(), out.numel(), torch.cuda.current_stream().cuda_stream), "dbg")
    return out.cpu().numpy().reshape(shape)


def run(H, L, B, gen, dtype, seed=3, tr16=1):
    p = vo.init_params(L, H, seed, gen)
    # make BN affine and biases non-trivial so every term is exercised
    rng = np.random.default_rng(seed)
    for k in p:
        if k.endswith(".1.weight"): p[k] = 1 + 0.2 * rng.standard_normal(p[k].shape)
        if k.endswith(".1.bias") or k.endswith(".0.bias") or k.endswith("3.bias"): p[k] = 0.1 * rng.standard_normal(p[k].shape)
    model = VanillaVAE(1, L, H, generalised=gen, compute_dtype=dtype).to("cuda")
    sd = model.state_dict()
    for k, v in p.items(): sd[k] = torch.from_numpy(v).float()
    model.load_state_dict(sd)
    x = vo.synth_pianoroll(B, H, seed)
    eps = vo.counter_normal(B * L, seed, 5).reshape(B, L)
    xt = torch.from_numpy(x).cuda(); et = torch.from_numpy(eps).float().cuda()
    model._context(B)
    _lib.lib().vae_set_option(model._ctx.handle, b"use_tr16", tr16)
    out3, xhat = model.fused_forward_backward(xt, eps=et)
    torch.cuda.synchronize()
    c = vo.forward(p, x.astype(np.float64), eps, None, train=True)
    lo = vo.loss(c); g = vo.backward(p, c)
    res = {"cfg": [H, L, B, gen, dtype, tr16]}
    names = vo._ENC + vo._DEC + ["final_layer"]
    for i, n in enumerate(names):
        res[f"y{i}"] = rel(dbg(model, i, c[n + ".y"].shape), c[n + ".y"])
    res["mu"] = rel(model._last["mu"].cpu().numpy(), c["mu"]); res["lv"] = rel(model._last["lv"].cpu().numpy(), c["lv"])
    res["z"] = rel(model._last["z"].cpu().numpy(), c["zlat"])
    s = int(round((c["zlat"] @ p["decoder_input.weight"].T).shape[1] // 256) ** 0.5)
    res["dd0"] = rel(dbg(model, 17, (B, 256, s, s)), g["__dd0"].reshape(B, 256, s, s)) if "__dd0" in g else None
    for i, n in enumerate(names):
        if n + ".dz" in g: res[f"dz{i}"] = rel(dbg(model, 8 + i, g[n + ".dz"].shape), g[n + ".dz"])
    res["d0"] = rel(dbg(model, 16, (B, 256, s, s)), (c["zlat"] @ p["decoder_input.weight"].T + p["decoder_input.bias"]).reshape(B, 256, s, s))
    res["xhat"] = rel(xhat.cpu().numpy(), c["output"])
    res["loss"] = [out3.tolist(), [float(lo["loss"]), float(lo["reconstruction_loss"]), float(lo["kld_loss"])]]
    gflat = model.flat_grads().cpu().numpy()
    for i, n in enumerate(_lib.PARAM_NAMES):
        got = gflat[model._offs[i]:model._offs[i] + model._sizes[i]].reshape(g[n].shape)
        res["g/" + n] = [rel(got, g[n]), float(np.sqrt((g[n] ** 2).sum()))]
    return res


if __name__ == "__main__":
    os.makedirs("gpurun_out", exist_ok=True)
    rc = _lib.lib().vae_selftest_tr16(torch.cuda.current_stream().cuda_stream)
    print("tr16 selftest rc", rc, _lib.lib().vae_last_error())
    allres = []
    import ast
    cfgs = ast.literal_eval(sys.argv[1]) if len(sys.argv) > 1 else [(32, 16, 4, False, "f32", 1), (32, 16, 33, False, "f32", 1), (64, 64, 8, True, "f32", 1),
                (32, 16, 4, False, "bf16", 0), (32, 16, 4, False, "bf16", 1), (64, 16, 4, True, "bf16", 1), (128, 16, 2, True, "bf16", 1)]
    for cfg in cfgs:
        try:
            r = run(*cfg[:5], tr16=cfg[5])
        except Exception as e:  # noqa
            r = {"cfg": list(cfg), "error": repr(e)}
        allres.append(r)
        print(json.dumps(r))
        sys.stdout.flush()
    json.dump(allres, open("gpurun_out/debug_layers.json", "w"), indent=1)
