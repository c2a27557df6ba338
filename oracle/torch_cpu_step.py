"""CPU port of the reference's training step on PyTorch CPU ops (TEST / BASELINE INFRASTRUCTURE ONLY).

A functional restatement of /root/reference/midi_autoencoder models.py:41-82,107-225 and
train.py:201-238,634-659 using the same ATen operators the reference dispatches (conv2d,
conv_transpose2d, batch_norm, leaky_relu, linear, sigmoid, binary_cross_entropy, AdamW,
OneCycleLR), so it is what "the reference's own CPU train.py" costs on a host.  It exists because
the reference itself cannot travel to the GPU box.  Pinned against the golden fixtures in
tests/test_oracle.py.  Used only by tests/ and by bench.py's cpu_baseline leg (kind "port");
the product path never imports it.
"""
from __future__ import annotations

import time

import numpy as np
import torch
import torch.nn.functional as F

from . import vae_oracle as vo


class TorchCpuStep:
    def __init__(self, params: dict, kld_weight=1.0, batch=32, total_steps=100, lr_relative=0.01, dtype=torch.float32):
        self.p = {k: torch.tensor(np.asarray(v), dtype=dtype, requires_grad=True) for k, v in params.items()}
        self.bn = {k: torch.tensor(np.asarray(v)) if k.endswith("num_batches_tracked") else torch.tensor(np.asarray(v), dtype=dtype)
                   for k, v in vo.init_bn_state().items()}
        self.kld_weight = kld_weight
        lr = vo.scaled_lr(lr_relative, batch)
        enc = [v for k, v in self.p.items() if k.startswith("encoder.")]
        dec = [v for k, v in self.p.items() if k.startswith("decoder.")]
        # train.py:210-228: only encoder and decoder are optimised
        self.opt = torch.optim.AdamW([{"params": enc, "lr": lr}, {"params": dec, "lr": lr}], lr=lr, weight_decay=0.0)
        self.sched = torch.optim.lr_scheduler.OneCycleLR(self.opt, [lr, lr], epochs=1, steps_per_epoch=total_steps)

    def _bn(self, name, y):
        return F.batch_norm(y, self.bn[name + ".running_mean"], self.bn[name + ".running_var"], self.p[name + ".weight"],
                            self.p[name + ".bias"], training=True, momentum=0.1, eps=1e-5)

    def forward(self, x, eps):
        p = self.p
        a = x
        for i in range(4):                                               # models.py:41-51,129
            a = F.leaky_relu(self._bn(f"encoder.{i}.1", F.conv2d(a, p[f"encoder.{i}.0.weight"], p[f"encoder.{i}.0.bias"], stride=2, padding=1)))
        pre = a.flatten(start_dim=1)                                     # models.py:133
        mu = F.linear(pre, p["fc_mu.weight"], p["fc_mu.bias"])           # :137
        lv = F.linear(pre, p["fc_var.weight"], p["fc_var.bias"])         # :141
        z = eps * torch.exp(0.5 * lv) + mu                               # :181-183
        d = F.linear(z, p["decoder_input.weight"], p["decoder_input.bias"])
        s = int(round((d.shape[1] // 256) ** 0.5))
        a = d.view(-1, 256, s, s)                                        # :166 (generalised s)
        for i in range(3):                                               # :62-73,167
            a = F.leaky_relu(self._bn(f"decoder.{i}.1", F.conv_transpose2d(a, p[f"decoder.{i}.0.weight"], p[f"decoder.{i}.0.bias"], stride=2, padding=1, output_padding=1)))
        a = F.leaky_relu(self._bn("final_layer.1", F.conv_transpose2d(a, p["final_layer.0.weight"], p["final_layer.0.bias"], stride=2, padding=1, output_padding=1)))
        xhat = torch.sigmoid(F.conv2d(a, p["final_layer.3.weight"], p["final_layer.3.bias"], stride=1, padding=1))
        return xhat, mu, lv, z

    def step(self, x, eps):
        """train.py:634-659: forward, zero_grad, loss, backward, optimizer.step, scheduler.step."""
        xhat, mu, lv, z = self.forward(x, eps)
        self.opt.zero_grad()
        recon = F.binary_cross_entropy(xhat, x)                          # models.py:208
        kld = -0.5 * torch.mean(torch.sum(1 + lv - mu ** 2 - torch.exp(lv), dim=-1))  # :214
        loss = recon + self.kld_weight * kld
        loss.backward()
        self.opt.step()
        self.sched.step()
        return float(loss.detach()), float(recon.detach()), float(-kld.detach())


def time_cpu_baseline(img_size, latent_dim, batch, generalised=True, budget_s=20.0, max_steps=50, threads=None, seed=0, warmup=1):
    """samples/s of the CPU port on a bounded sample of the bench workload."""
    if threads:
        torch.set_num_threads(threads)
    params = vo.init_params(latent_dim, img_size, seed, generalised)
    st = TorchCpuStep(params, batch=batch, total_steps=max_steps + max(1, warmup) + 1)
    x = torch.from_numpy(vo.synth_pianoroll(batch, img_size, seed))
    eps = torch.from_numpy(vo.counter_normal(batch * latent_dim, seed, 5).reshape(batch, latent_dim)).float()
    for _ in range(max(1, warmup)):
        st.step(x, eps)  # warm-up
    t0 = time.perf_counter()
    n = 0
    while n < max_steps and (time.perf_counter() - t0) < budget_s:
        st.step(x, eps)
        n += 1
    dt = time.perf_counter() - t0
    return {"value": batch * n / dt, "steps": n, "seconds": dt, "threads": torch.get_num_threads()}
