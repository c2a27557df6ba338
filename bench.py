#!/usr/bin/env python3
"""Benchmark of the MI355X-native VAE training step (the metric BASELINE.json names).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one pass of the hot path over one resident synthetic batch: forward, ELBO,
backward, [gradient all-reduce over RCCL when N>1], fused AdamW, OneCycle scheduler step
(train.py:634-659 of the reference).  Inputs are generated on the device before the timed
region (H2D excluded).  Workload at N=1: BASELINE.json configs[1] on the metric's 128x128
pianoroll: VanillaVAE latent_dim=16, batch 256, bf16 storage / f32 accumulate (generalised
model).  Rank 0 prints ONE JSON line.
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# main stream + three side streams (+ the communication stream, + a process group's own): more than HIP's default of 4
# hardware queues, and streams that share a queue serialise (measured with a process group present: 1.47 ms/step at 4
# queues, 1.39 at 6..8; without one 4..8 are the same).  Set before HIP starts.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "6" if os.environ.get("VAE_DP_OVERLAP") == "1" else "8")   # (see torch_vae_amd/__init__.py)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec (MI355X_MICROARCH.md: 8.0 TB/s spec, 6.29 TB/s measured copy)
MFMA_PEAK_TF = {"bf16": 2500.0, "f16": 2500.0, "f32": 157.3}
ESZ = {"bf16": 2, "f16": 2, "f32": 4}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--size", type=int, default=128, help="pianoroll side (128 = the metric's; 32 = reference-exact model)")
    ap.add_argument("--batch", type=int, default=256, help="per-GPU batch")
    ap.add_argument("--latent", type=int, default=16)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f16", "f32"])
    ap.add_argument("--no-extras", action="store_true", help="skip the extra records (f32 mode, 32x32 model, drop-in loop, ELBO gap)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-budget", type=float, default=15.0)
    ap.add_argument("--kernels", action="store_true", help="print the per-kernel table to stderr")
    ap.add_argument("--force-dist", action="store_true", help="initialise torch.distributed (RCCL) even with one rank (path test)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="nccl = RCCL (the product path); gloo only rehearses the multi-rank control flow with ranks sharing one GPU")
    ap.add_argument("--set", action="append", default=[], metavar="KNOB=VALUE", help="vae_set_option knob (diagnostics)")
    ap.add_argument("--dump-order", default=None, help="write the per-step launch order (label, kernel symbol) as JSON")
    return ap.parse_args()


def host_cores():
    """Host cores this process may really use: the scheduler affinity mask, capped by the cgroup CPU quota when there is one (a GPU
    box hands a one-GPU job a share of a large host: threads beyond the quota only fight each other).  BENCH_CPU_THREADS overrides."""
    if os.environ.get("BENCH_CPU_THREADS"):
        return max(1, int(os.environ["BENCH_CPU_THREADS"])), "BENCH_CPU_THREADS"
    n = os.cpu_count() or 1
    how = "os.cpu_count"
    try:
        n = len(os.sched_getaffinity(0)); how = "sched_getaffinity"
    except Exception:
        pass
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                quota, period = txt[0], float(txt[1])
            else:
                quota, period = txt[0], float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if quota not in ("max", "-1") and float(quota) > 0:
                q = max(1, int(float(quota) / period + 0.5))
                if q < n:
                    n, how = q, f"cgroup cpu quota ({path})"
            break
        except Exception:
            continue
    return n, how


def timed_steps(step_fn, steps, warmup):
    for i in range(warmup):
        step_fn(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        step_fn(warmup + i)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps


def extra_records(args, dev, model, batches, Namespace, VanillaVAE, SyntheticPianorollLoader, build_optimizer, fused_step,
                  train_one_epoch, algorithmic_bytes_per_step, count_flops_per_sample):
    """Records beside the headline number (rank 0, one GPU, outside the timed region):
      elbo_rel_gap  the benched storage mode against the parity-proven f32 mode on the same weights, batch and noise
      f32           the same workload in the f32 kernel mode (the configuration that meets the 1e-4 ELBO target)
      reference_exact_32x32   SURVEY.md 8(d) config 2 at the reference's own 32x32 model
      train_one_epoch_samples_per_s   the drop-in loop itself over host-resident BIT-PLANE batches (0.5 MB copy, expansion on the device,
                                      the loop's loss bookkeeping); train_one_epoch_{uint8,f32}_host_samples_per_s: byte / float32 host batches"""
    H, L, B = args.size, args.latent, args.batch
    gen = H != 32
    out = {}

    def make(dtype, H_, B_):
        m = VanillaVAE(1, L, H_, generalised=H_ != 32, compute_dtype=dtype, max_batch=B_).to(dev)
        cfg = Namespace(batch_size_per_gpu=B_, world_size=1, lr_relative=0.01, weight_decay=0.0, optimizer="AdamW", scheduler="OneCycle",
                        epochs=1, freeze_encoder=False, log_wandb=False, print_interval=10 ** 9, log_interval=10 ** 9, global_rank=1)
        o, s_ = build_optimizer(cfg, m, steps_per_epoch=100000)
        return m, o, s_, cfg

    if args.dtype != "f32":
        m32, o32, s32, _ = make("f32", H, B)
        m32.flat_parameters().copy_(model.flat_parameters()); m32._bnflat.copy_(model._bnflat)
        eps = torch.randn(B, L, device=dev)
        a3, _ = model.fused_forward_backward(batches[0], eps=eps)
        b3, _ = m32.fused_forward_backward(batches[0], eps=eps)
        a3, b3 = a3.tolist(), b3.tolist()
        gaps = {"loss": abs(a3[0] / b3[0] - 1), "reconstruction_loss": abs(a3[1] / b3[1] - 1), "kld_loss": abs(a3[2] / b3[2] - 1)}
        out["elbo_rel_gap"] = {"vs": "f32 kernel mode (<= 1e-4 of the reference), same weights / batch / eps", **gaps,
                               "target": 1e-4, "meets_1e-4": bool(max(gaps.values()) <= 1e-4),
                               "note": f"the benched {args.dtype} storage mode; the f32 kernel mode (record 'f32') is the one that meets 1e-4"}

        def step32(i):
            fused_step(m32, o32, batches[i % 4]); s32.step()
        sec = timed_steps(step32, max(5, args.steps // 2), 3)
        by, fl = algorithmic_bytes_per_step(H, L, B, 4, gen), count_flops_per_sample(H, L, gen) * B
        out["f32"] = {"ms_per_step": round(1e3 * sec, 4), "samples_per_s": round(B / sec, 1),
                      "hbm_frac": round(by / sec / 1e9 / HBM_PEAK_GBS, 4), "mfma_frac": round(fl / sec / 1e12 / MFMA_PEAK_TF["f32"], 4)}
        del m32, o32, s32
    if H != 32:
        mr, orr, sr, _ = make(args.dtype, 32, B)
        xr = [SyntheticPianorollLoader(B, 32, n_batches=1, seed=50 + i, device=dev).batch(0)[0] for i in range(4)]

        def stepr(i):
            fused_step(mr, orr, xr[i % 4]); sr.step()
        sec = timed_steps(stepr, 2 * args.steps, 5)
        by = algorithmic_bytes_per_step(32, L, B, ESZ[args.dtype], False)
        out["reference_exact_32x32"] = {"workload": f"VanillaVAE(1, {L}, 32) as the reference builds it (models.py:33,166), batch {B}, {args.dtype}",
                                        "ms_per_step": round(1e3 * sec, 4), "samples_per_s": round(B / sec, 1),
                                        "hbm_frac": round(by / sec / 1e9 / HBM_PEAK_GBS, 4)}
        del mr, orr, sr
    # the drop-in loop (train.py:554-767 mirror) over host-resident batches, as a DataLoader would hand them over
    ml, ol, sl, cfgl = make(args.dtype, H, B)
    nb = max(8, args.steps)
    host = [(batches[i % 4].cpu().pin_memory(), torch.zeros(B, dtype=torch.long)) for i in range(4)]
    loader = [host[i % 4] for i in range(nb)]
    train_one_epoch(cfgl, ml, ol, sl, ml.loss, loader[:4], device=dev, epoch=2)      # warm-up
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    train_one_epoch(cfgl, ml, ol, sl, ml.loss, loader, device=dev, epoch=2)
    torch.cuda.synchronize()
    out["train_one_epoch_f32_host_samples_per_s"] = round(nb * B / (time.perf_counter() - t0), 1)
    # the same loop fed uint8 pianorolls (the cells are 0/1: INTEGRATION.md shows the one-line collate change): 4 MB blocking copy,
    # expanded to float32 on the device
    host8 = [(h[0].to(torch.uint8).pin_memory(), h[1]) for h in host]
    loader8 = [host8[i % 4] for i in range(nb)]
    train_one_epoch(cfgl, ml, ol, sl, ml.loss, loader8[:min(nb, 12)], device=dev, epoch=2)     # warm-up (the copy stream's allocator blocks, the pinned staging: the first epoch over a new loader is 10 % slower)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    train_one_epoch(cfgl, ml, ol, sl, ml.loss, loader8, device=dev, epoch=2)
    torch.cuda.synchronize()
    out["train_one_epoch_uint8_host_samples_per_s"] = round(nb * B / (time.perf_counter() - t0), 1)
    # ... and bit planes (torch_vae_amd.train.pack_bits: 0.5 MB per batch), expanded on the device
    from torch_vae_amd.train import pack_bits
    hostb = [(pack_bits(h[0]).pin_memory(), h[1]) for h in host]
    loaderb = [hostb[i % 4] for i in range(nb)]
    train_one_epoch(cfgl, ml, ol, sl, ml.loss, loaderb[:min(nb, 12)], device=dev, epoch=2)     # warm-up
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    train_one_epoch(cfgl, ml, ol, sl, ml.loss, loaderb, device=dev, epoch=2)
    torch.cuda.synchronize()
    out["train_one_epoch_samples_per_s"] = round(nb * B / (time.perf_counter() - t0), 1)
    return out


def main():
    args = parse()
    # Native libraries (RCCL prints a version banner) write to fd 1; keep stdout for the ONE JSON line.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    rank = int(os.environ.get("RANK", 0))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the VAE step has no CPU path")
    if args.backend == "gloo":   # rehearsal of the N>1 control flow on a one-GPU box: ranks share the card
        local_rank %= torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1 or args.force_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
        if args.backend == "gloo":
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    from argparse import Namespace
    from torch_vae_amd import _lib
    from torch_vae_amd.models import VanillaVAE, algorithmic_bytes_per_step, count_flops_per_sample
    from torch_vae_amd.train import SyntheticPianorollLoader, build_optimizer, enable_library_allreduce, fused_step, train_one_epoch

    H, L, B = args.size, args.latent, args.batch
    gen = H != 32
    torch.manual_seed(0)  # identical initial weights on every rank
    model = VanillaVAE(1, L, H, generalised=gen, compute_dtype=args.dtype, max_batch=B).to(dev)
    for kv in args.set:
        k, v = kv.split("=")
        _lib.check(_lib.lib().vae_set_option(model._context(B).handle, k.encode(), int(v)), "vae_set_option")
    total_steps = args.steps + args.warmup + 8
    cfg = Namespace(batch_size_per_gpu=B, world_size=world, lr_relative=0.01, weight_decay=0.0, optimizer="AdamW",
                    scheduler="OneCycle", epochs=1, freeze_encoder=False)
    opt, sched = build_optimizer(cfg, model, steps_per_epoch=total_steps)   # world > 1: also broadcasts rank 0's state, enables the library's RCCL exchange
    if args.force_dist and args.backend == "nccl":
        enable_library_allreduce(model)   # single-rank path test of vae_comm_init / vae_allreduce_grads
    pool = SyntheticPianorollLoader(B, H, n_batches=4, seed=1000 * rank, device=dev, pool=4)
    batches = [pool.batch(i)[0] for i in range(4)]
    model.eps_seed = 7919   # (the rank is mixed into the 64-bit seed by the model: replicas draw independent noise)
    losses = torch.zeros(3, device=dev)

    def step(i):
        out3, _ = fused_step(model, opt, batches[i % 4])
        sched.step()
        return out3

    for i in range(args.warmup):
        losses = step(i)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        losses = step(args.warmup + i)
    host_enqueue_ms = 1e3 * (time.perf_counter() - t0) / args.steps   # host time to enqueue a step (diagnostic: < ms_per_step when the GPU is the limit)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    final_loss = losses.tolist()

    # per-kernel durations (HIP events on each launch stream), outside the timed region.  Weight gradients run on
    # side streams concurrently with the input-gradient chain, so a launch's duration here (and in rocprofv3's
    # trace of this same command) includes the time it shares the GPU; the roofline object therefore also carries
    # the throughput of everything running inside the dominant launch's window, and the same launch measured
    # alone (side streams off).
    roofline, kernels = None, []
    L_ = _lib.lib()
    h = model._ctx.handle
    nprof = 3

    def profile_steps():
        # EVERY rank runs these steps (each one contains the gradient all-reduce); only rank 0 records and reads events
        if rank == 0:
            _lib.check(L_.vae_profile(h, 1), "vae_profile")
        for i in range(nprof):
            step(args.warmup + args.steps + i)
        torch.cuda.synchronize()
        if rank != 0:
            return None, None, None
        buf = ctypes.create_string_buffer(1 << 16)
        _lib.check(L_.vae_profile_report(h, buf, len(buf)), "vae_profile_report")
        tbuf = ctypes.create_string_buffer(1 << 20)
        _lib.check(L_.vae_profile_timeline(h, tbuf, len(tbuf)), "vae_profile_timeline")
        seq = None
        if args.dump_order:
            sbuf = ctypes.create_string_buffer(1 << 18)
            _lib.check(L_.vae_profile_sequence(h, sbuf, len(sbuf)), "vae_profile_sequence")
            seq = json.loads(sbuf.value.decode())
        _lib.check(L_.vae_profile(h, 0), "vae_profile")
        ks = json.loads(buf.value.decode())
        for k in ks:
            k["ms_per_call"] = k["ms"] / k["calls"]
            k["gbs"] = k["bytes"] / k["ms"] / 1e6 if k["ms"] > 0 else 0.0
            k["tflops"] = k["flops"] / k["ms"] / 1e9 if k["ms"] > 0 else 0.0
        ks.sort(key=lambda k: -k["ms"])
        return ks, json.loads(tbuf.value.decode()), seq

    kernels, timeline, seq = profile_steps()
    # the same launches alone: side streams off for a few profiled steps (again on every rank)
    iso = None
    if not any(kv.startswith("use_side_stream=") for kv in args.set):
        _lib.check(L_.vae_set_option(h, b"use_side_stream", 0), "vae_set_option")
        step(0); torch.cuda.synchronize()
        iso, _, _ = profile_steps()
        _lib.check(L_.vae_set_option(h, b"use_side_stream", 1), "vae_set_option")
    if rank != 0:
        kernels = []
    else:
        if args.dump_order and seq is not None:
            json.dump(seq[:len(seq) // nprof], open(args.dump_order, "w"))
        # dominant = most total time among the launches that move real data (>= 10 % of the largest per-launch byte count):
        # bookkeeping launches (reductions of a few KB, finalisations) can take long when ranks share a GPU, yet say nothing
        bmax = max(k["bytes"] / k["calls"] for k in kernels)
        dom = next((k for k in kernels if k["bytes"] / k["calls"] >= 0.1 * bmax), kernels[0])
        ach = dom["gbs"]
        # algorithmic bytes of every launch overlapping the dominant launch's windows / total window time
        wbytes, wtime, mates = 0.0, 0.0, set()
        for (n0, s0, e0, _b) in timeline:
            if n0 != dom["name"]:
                continue
            wtime += e0 - s0
            for (n1, s1, e1, b1) in timeline:
                ov = min(e0, e1) - max(s0, s1)
                if ov > 0 and e1 > s1:
                    wbytes += b1 * ov / (e1 - s1)
                    if n1 != n0 and b1 * ov / (e1 - s1) > 0.02 * dom["bytes"] / dom["calls"]:
                        mates.add(n1)
        window_gbs = wbytes / wtime / 1e6 if wtime > 0 else 0.0
        isolated = None
        for k in iso or []:
            if k["name"] == dom["name"]:
                isolated = {"avg_launch_us": round(1e3 * k["ms_per_call"], 2), "achieved": round(k["gbs"], 1),
                            "frac": round(k["gbs"] / HBM_PEAK_GBS, 4)}
        # HBM bytes per launch of that kernel from the rocprofv3 PMC passes (FETCH_SIZE x2 + WRITE_SIZE, separate
        # passes; tools/pmc_traffic.py writes profiles/pmc_traffic.json on the GPU box) - null when not collected
        traffic = None
        try:
            tj = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
            key = f"{dom['name']}|H{H}|L{L}|B{B}|{args.dtype}"
            if key in tj:
                traffic = tj[key]["traffic_bytes_per_launch"]
        except Exception:
            traffic = None
        # the same figures for the dominant launch on the CALLER's stream (the critical chain); the overall dominant launch
        # is usually a weight gradient that is deliberately throttled on a side stream
        crit = next((k for k in kernels if not k.get("side")), dom)
        critical = {"kernel": crit["name"], "avg_launch_us": round(1e3 * crit["ms_per_call"], 2), "achieved": round(crit["gbs"], 1),
                    "frac": round(crit["gbs"] / HBM_PEAK_GBS, 4), "algorithmic_bytes_per_launch": crit["bytes"] / crit["calls"]}
        roofline = {"bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": traffic,
                    "traffic_source": None if traffic is None else "profiles/pmc_traffic.json (rocprofv3 PMC passes of this command, collected on the builder's GPU box and replayed here, not measured in this run)",
                    "kernel": dom["name"], "on_side_stream": bool(dom.get("side")), "critical_stream_dominant": critical,
                    "avg_launch_us": round(1e3 * dom["ms_per_call"], 2),
                    "algorithmic_bytes_per_launch": dom["bytes"] / dom["calls"],
                    "kernel_tflops": round(dom["tflops"], 1),
                    "concurrent_window": {"achieved": round(window_gbs, 1), "frac": round(window_gbs / HBM_PEAK_GBS, 4),
                                          "with": sorted(mates)},
                    "isolated": isolated}
        # the launch label with the most total GPU time per step, whatever it moves (split-K reductions and other bookkeeping
        # launches included): the class of cost the byte filter above hides
        top = kernels[0]
        roofline["top_by_time"] = {"kernel": top["name"], "calls_per_step": top["calls"] // nprof, "us_per_step": round(1e3 * top["ms"] / nprof, 2),
                                   "achieved": round(top["gbs"], 1), "frac": round(top["gbs"] / HBM_PEAK_GBS, 4)}
        byname = {}
        for k in kernels:   # labels carry the layer tag ("reduce_slab @encoder.2"): totals per kernel family
            fam = k["name"].split(" @")[0]
            byname.setdefault(fam, [0, 0.0]); byname[fam][0] += k["calls"]; byname[fam][1] += k["ms"]
        fam, (fc, fms) = max(byname.items(), key=lambda kv: kv[1][1])
        roofline["top_family_by_time"] = {"kernel": fam, "launches_per_step": fc // nprof, "us_per_step": round(1e3 * fms / nprof, 2)}
        if args.kernels:
            tot = sum(k["ms"] for k in kernels) / nprof
            print(f"per-step kernel time {tot:.3f} ms (sum over streams)", file=sys.stderr)
            for k in kernels:
                print(f"  {k['name']:52s} calls/step {k['calls'] // nprof:3d}  {k['ms'] / nprof:8.3f} ms/step  {k['gbs']:8.1f} GB/s  {k['tflops']:7.1f} TF", file=sys.stderr)

    cpu = cpu_config0 = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle.torch_cpu_step import time_cpu_baseline
        ncores, cores_how = host_cores()
        print(f"bench.py: cpu baseline on {ncores} host threads ({cores_how})", file=sys.stderr, flush=True)
        cb = min(B, 32) if H >= 128 else B
        r = time_cpu_baseline(H, L, cb, generalised=gen, budget_s=args.cpu_budget, threads=ncores)
        cpu = {"value": round(r["value"], 1), "unit": "samples/s", "cores": r["threads"], "cores_from": cores_how, "kind": "port",
               "sample": f"{r['steps']} steps of the same model/input size at batch {cb}, f32, torch CPU ops "
                         f"(oracle/torch_cpu_step.py), {r['seconds']:.1f} s"}
        print(f"bench.py: cpu baseline {cpu['value']} samples/s; config0 next", file=sys.stderr, flush=True)
        # BASELINE.json configs[0] / BASELINE.md section 3: the reference-exact 32x32 model, batch 32, f32, 200 timed steps after
        # 5 warm-up steps on all host cores; and batch 256 beside it (like for like with configs[1])
        r0 = time_cpu_baseline(32, L, 32, generalised=False, budget_s=40.0, max_steps=200, threads=ncores, warmup=5)
        print(f"bench.py: config0 batch 32: {r0['steps']} steps in {r0['seconds']:.1f} s", file=sys.stderr, flush=True)
        r1 = time_cpu_baseline(32, L, 256, generalised=False, budget_s=20.0, max_steps=40, threads=ncores, warmup=5)
        cpu_config0 = {"value": round(r0["value"], 1), "unit": "samples/s", "cores": r0["threads"], "kind": "port",
                       "sample": f"{r0['steps']} steps after 5 warm-up, VanillaVAE(1, {L}, 32) reference-exact, batch 32, f32, {r0['seconds']:.1f} s",
                       "batch_256": {"value": round(r1["value"], 1), "sample": f"{r1['steps']} steps after 5 warm-up, batch 256, {r1['seconds']:.1f} s"}}

    extras = {}
    if rank == 0:
        print("bench.py: timed region and profiles done", file=sys.stderr, flush=True)
    if rank == 0 and world == 1 and not args.no_extras:
        extras = extra_records(args, dev, model, batches, Namespace, VanillaVAE, SyntheticPianorollLoader, build_optimizer,
                               fused_step, train_one_epoch, algorithmic_bytes_per_step, count_flops_per_sample)
    if rank == 0:
        esz = ESZ[args.dtype]
        value = world * B * args.steps / dt
        step_bytes = algorithmic_bytes_per_step(H, L, B, esz, gen)
        step_flops = count_flops_per_sample(H, L, gen) * B
        ms = 1e3 * dt / args.steps
        out = {
            "metric": "training samples/sec (VAE step: forward+ELBO+backward+AdamW), synthetic pianoroll",
            "value": round(value, 1), "unit": "samples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms, 4), "host_enqueue_ms_per_step": round(host_enqueue_ms, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": f"VanillaVAE latent_dim={L}, {H}x{H} synthetic pianoroll, batch {B}/GPU"
                                   f"{' (generalised bottleneck 256*(H/16)^2)' if gen else ' (reference-exact 32x32 model)'}",
                       "img_size": H, "latent_dim": L, "batch_per_gpu": B, "global_batch": B * world,
                       "parallelism": f"dp{world}", "optimizer": "AdamW+OneCycle (encoder, decoder groups)"},
            "roofline": roofline,
            "step_roofline": {"algorithmic_bytes": step_bytes, "algorithmic_flops": step_flops,
                              "hbm_frac": round(step_bytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                              "mfma_frac": round(step_flops / (ms * 1e-3) / 1e12 / MFMA_PEAK_TF[args.dtype], 4)},
            "cpu_baseline": cpu, "cpu_baseline_config0": cpu_config0,
            "elbo_last_step": {"loss": final_loss[0], "reconstruction_loss": final_loss[1], "kld_loss": final_loss[2]},
        }
        if world > 1 or args.force_dist:
            out["exchange"] = ("library RCCL communicator (vae_allreduce_grads)" if model.library_comm_world() == max(world, 1)
                               else f"torch.distributed ({args.backend})")
            out["rccl_ranks"] = int(model.library_comm_world())   # ranks of the step library's own RCCL communicator (vae_comm_world)
        out.update(extras)
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    rccl_ok = True
    if (world > 1 or args.force_dist) and args.backend == "nccl":
        rccl_ok = int(model.library_comm_world()) == world   # the product path: every rank inside ONE RCCL communicator
        if not rccl_ok:
            print(f"bench.py: rank {rank}: the step library's RCCL communicator has {model.library_comm_world()} ranks, expected {world}", file=sys.stderr)
    if dist.is_initialized():
        dist.destroy_process_group()
    if not rccl_ok:
        raise SystemExit(3)


if __name__ == "__main__":
    main()
