"""One steady-state step of `bench.py` from a rocprofv3 kernel trace: start / end / queue of every launch.

    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tr -- python3 bench.py --steps 10 --warmup 5 --no-cpu-baseline --no-extras
    python tools/trace_timeline.py gpurun_out/tr [--step 8] > profiles/r02_timeline.txt

A step is delimited by the AdamW launches (one per step).  Columns: start and end in us relative to the step's first
launch, duration, queue, the gap to the previous launch of the SAME queue, short kernel name.
"""
import csv, glob, os, re, sys


def short(name):
    name = re.sub(r"\(.*", "", name)
    name = re.sub(r"^void ", "", name)
    return name[:90]


def main():
    root = sys.argv[1]
    want = int(sys.argv[sys.argv.index("--step") + 1]) if "--step" in sys.argv else -3
    f = sorted(glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True))[0]
    rows = [r for r in csv.DictReader(open(f)) if r["Kind"] == "KERNEL_DISPATCH"]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    cuts = [i for i, r in enumerate(rows) if "adamw" in r["Kernel_Name"].lower()]
    if len(cuts) < 4:
        sys.exit("fewer than 4 optimiser launches in the trace")
    a, b = cuts[want - 1] + 1, cuts[want] + 1
    step = rows[a:b]
    t0 = int(step[0]["Start_Timestamp"])
    last_end = {}
    busy = {}
    print(f"# launches {len(step)}; step span {(int(step[-1]['End_Timestamp']) - t0) / 1e3:.1f} us; "
          f"previous step's AdamW ended {(t0 - int(rows[a - 1]['End_Timestamp'])) / 1e3:.1f} us before the first launch")
    for r in step:
        s, e, q = int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Queue_Id"]
        gap = (s - last_end[q]) / 1e3 if q in last_end else float("nan")
        last_end[q] = e
        busy[q] = busy.get(q, 0) + (e - s)
        print(f"{(s - t0) / 1e3:8.1f} {(e - t0) / 1e3:8.1f} {(e - s) / 1e3:7.1f} us  q{q:>2}  gap {gap:6.1f}  {short(r['Kernel_Name'])}")
    for q, v in sorted(busy.items()):
        print(f"# queue {q}: busy {v / 1e3:.1f} us")


if __name__ == "__main__":
    main()
