#!/usr/bin/env python3
"""SQ-counter fractions per kernel from the two passes of tools/make_pmc_sq.sh.
    python tools/pmc_sq_table.py gpurun_out/pmc_sq_v2 > profiles/r02_pmc_sq_v2.txt
Fractions are of SQ_WAVE_CYCLES (wave-resident cycles); valu/mfma = VALU instructions per MFMA instruction."""
import collections, csv, glob, os, re, sys


def short(n):
    n = re.sub(r"\(.*", "", n).replace("void ", "")
    return n[:90]


acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        acc[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))


def avg(v, k):
    return sum(v[k]) / len(v[k]) if v.get(k) else 0.0


print("# SQ counters per kernel, bench.py workload (H=128 L=16 B=256 bf16), side streams off, two rocprofv3 --pmc passes (tools/make_pmc_sq.sh)")
print("kernel | wait_any | wait_inst (issue stall) | active_any | active_valu | valu/mfma | lds_bank_conflict/lds_idx_active | dispatches")
for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1].get("SQ_WAVE_CYCLES", [0]))):
    wc = avg(v, "SQ_WAVE_CYCLES")
    if wc <= 0:
        continue
    mf, lds = avg(v, "SQ_INSTS_MFMA"), avg(v, "SQ_LDS_IDX_ACTIVE")
    print(f"{k} | {avg(v, 'SQ_WAIT_ANY') / wc:.2f} | {avg(v, 'SQ_WAIT_INST_ANY') / wc:.2f} | {avg(v, 'SQ_ACTIVE_INST_ANY') / wc:.2f} | "
          f"{avg(v, 'SQ_ACTIVE_INST_VALU') / wc:.2f} | " + (f"{avg(v, 'SQ_INSTS_VALU') / mf:.0f}" if mf > 0 else "-") + " | "
          + (f"{avg(v, 'SQ_LDS_BANK_CONFLICT') / lds:.2f}" if lds > 0 else "-") + f" | {max(len(x) for x in v.values())}")
