#!/usr/bin/env python3
"""Turn rocprofv3 PMC passes of bench.py into profiles/pmc_traffic.json (HBM bytes per launch per bench label).

Usage (on the GPU box, after `bench.py --kernels --dump-order gpurun_out/order.json` and the PMC passes):
    python tools/pmc_traffic.py <fetch_pass_dir> <write_pass_dir> <order.json> <H> <L> <B> <dtype> [--out file.json]

The two passes are separate rocprofv3 runs (`--pmc FETCH_SIZE --kernel-trace` and `--pmc WRITE_SIZE
--kernel-trace`, MI355X_MICROARCH.md HBM section).  FETCH_SIZE is doubled (gfx950 tallies 128-byte requests
at 64 B); both counters are in KiB.  Dispatches are matched to bench labels by launch order inside a step:
order.json lists, for one step, the label of every profiled launch of the C ABI in order.
"""
import csv, glob, json, os, re, sys, collections


def load(d, counter):
    f = (glob.glob(os.path.join(d, "*counter_collection.csv")) + glob.glob(os.path.join(d, "*", "*counter_collection.csv")))[0]
    rows = [r for r in csv.DictReader(open(f)) if r["Counter_Name"] == counter]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    return rows


# label prefix -> regex of the device kernel(s) that label launches first
KERNEL_OF = [("convT_bwd_fused", r"convt_bwd_fused_kernel"), ("conv_bwd_fused", r"(?<![a-z_])conv_bwd_fused_kernel"), ("down_", r"(?<![a-z_])(down2?|dn3|dnfirst_stream)_kernel"), ("up_", r"(?<![a-z_])(up2?|up3|upfinal_stream)_kernel"), ("wgrad_kernel", r"(?<![a-z_])wgrad(_split)?_kernel"),
             ("convout_step", r"convout_(step|stream)"), ("convout_fwd", r"convout_fwd"), ("convout_bwd", r"convout_bwd"), ("conv1_fwd", r"conv1_fwd"),
             ("conv1_wgrad", r"conv1_wgrad"), ("dense", r"dense_kernel"), ("decin_fwd", r"decin_fwd|row_gemm_kernel"),
             ("decin_wgrad", r"decin_wgrad|batch_gemm_kernel"), ("fc_wgrad", r"fc_wgrad|batch_gemm_kernel"), ("fc_dgrad", r"fc_dgrad|row_gemm_kernel"), ("pack_weights", r"pack_kernel"),
             ("reduce_slab", r"reduce_slab_kernel"), ("bn_", r"bn_finalize_kernel")]


def regex_of(label):
    for pre, rx in KERNEL_OF:
        if label.startswith(pre):
            return re.compile(rx)
    raise KeyError(label)


def per_label(rows, labels):
    """Walk the dispatch stream; a step is matched when every label's kernel appears in order."""
    rxs = [regex_of(l) for l in labels]
    anyrx = re.compile("|".join(rx for _, rx in KERNEL_OF))
    seq = [r for r in rows if anyrx.search(r["Kernel_Name"])]
    out = collections.defaultdict(list)
    i = 0
    while i + len(labels) <= len(seq):
        window = seq[i:i + len(labels)]
        if all(rxs[k].search(window[k]["Kernel_Name"]) for k in range(len(labels))):
            for k, r in enumerate(window):
                out[labels[k]].append(float(r["Counter_Value"]))
            i += len(labels)
        else:
            i += 1
    return out


def main():
    fdir, wdir, order_json, H, L, B, dtype = sys.argv[1:8]
    out_path = sys.argv[sys.argv.index("--out") + 1] if "--out" in sys.argv else None
    order = json.load(open(order_json))
    fetch = per_label(load(fdir, "FETCH_SIZE"), order)
    write = per_label(load(wdir, "WRITE_SIZE"), order)
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "pmc_traffic.json")
    try:
        res = json.load(open(path))
    except Exception:
        res = {}
    suffix = f"|H{H}|L{L}|B{B}|{dtype}"
    res = {k: v for k, v in res.items() if not k.endswith(suffix)}   # this configuration is re-measured: drop labels that no longer exist
    # (figures are per device launch: a two-level slab reduction is two launches under one bench label)
    for label in fetch:
        if label not in write or not fetch[label]:
            continue
        f = 2.0 * 1024 * sum(fetch[label]) / len(fetch[label])
        w = 1024.0 * sum(write[label]) / len(write[label])
        res[f"{label}|H{H}|L{L}|B{B}|{dtype}"] = {"fetch_bytes_x2": f, "write_bytes": w, "traffic_bytes_per_launch": f + w,
                                                 "launches_averaged": len(fetch[label])}
    path = out_path or path
    json.dump(res, open(path, "w"), indent=1, sort_keys=True)
    print("wrote", path, len(res), "entries")


if __name__ == "__main__":
    main()
