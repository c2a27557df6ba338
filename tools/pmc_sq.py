#!/usr/bin/env python3
"""Per-kernel averages of rocprofv3 PMC counters (one or more pass directories), keyed by a short kernel name.
Usage: python tools/pmc_sq.py <pass_dir> [<pass_dir> ...]   (each from `rocprofv3 --pmc ... --kernel-trace --output-format csv`)"""
import collections, csv, glob, os, re, sys


def short(n):
    n = re.sub(r"\(.*", "", n)
    n = n.replace("void ", "")
    return n[:70]


def main():
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for d in sys.argv[1:]:
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                acc[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    names = sorted({c for k in acc.values() for c in k})
    print("kernel," + ",".join(names) + ",dispatches")
    for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1].get("SQ_WAVE_CYCLES", kv[1].get("SQ_BUSY_CYCLES", [0])))):
        n = max(len(x) for x in v.values())
        print(k + "," + ",".join(f"{sum(v[c]) / max(len(v[c]), 1):.0f}" if c in v else "" for c in names) + f",{n}")


if __name__ == "__main__":
    main()
