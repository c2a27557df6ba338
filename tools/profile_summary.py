#!/usr/bin/env python3
"""Write profiles/<tag>_summary.md from the artifacts of one profiling session.

    python tools/profile_summary.py <tag> <kernel_stats.csv> <bench.json> <per_kernel_hip_events.txt> <H> <L> <B> <dtype>

Inputs: rocprofv3 --kernel-trace --stats CSV of bench.py, bench.py's JSON line, bench.py --kernels table (HIP
events), profiles/pmc_traffic.json (tools/pmc_traffic.py).  Pure formatting; no numbers are derived here
except ratios of the listed columns.
"""
import csv, json, os, re, sys

tag, stats_csv, bench_json, events_txt, H, L, B, dtype = sys.argv[1:9]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
bench = json.loads(open(bench_json).read().strip().splitlines()[-1])
traffic = json.load(open(os.path.join(root, "profiles", "pmc_traffic.json")))
rows = list(csv.DictReader(open(stats_csv)))
ev = []
for l in open(events_txt):
    m = re.match(r"\s+(.+?)\s+calls/step\s+(\d+)\s+([\d.]+) ms/step\s+([\d.]+) GB/s\s+([\d.]+) TF", l)
    if m:
        ev.append((m.group(1), int(m.group(2)), float(m.group(3)), float(m.group(4)), float(m.group(5))))
out = [f"# Profile summary ({tag}) - bench.py, H={H} L={L} B={B} {dtype}, 1x MI355X", ""]
out += [f"Step: **{bench['ms_per_step']} ms**, {bench['value']} {bench['unit']}; step_roofline {bench['step_roofline']}.", ""]
r = bench["roofline"]
out += ["## Dominant launch (bench.py `roofline`)", "",
        f"* kernel: `{r['kernel']}`, {r['avg_launch_us']} us per launch while sharing the GPU with {r['concurrent_window']['with']}",
        f"* algorithmic bytes per launch {r['algorithmic_bytes_per_launch'] / 1e6:.1f} MB -> {r['achieved']} GB/s = {r['frac']} of {r['peak']} GB/s",
        f"* HBM-side traffic per launch (PMC, FETCH_SIZE x2 + WRITE_SIZE) {r['traffic'] / 1e6 if r['traffic'] else float('nan'):.1f} MB",
        f"* everything running inside its window: {r['concurrent_window']['achieved']} GB/s algorithmic = {r['concurrent_window']['frac']} of peak",
        f"* the same launch alone (side streams off): {r['isolated']}", ""]
out += ["## Top kernels by total time (rocprofv3 --kernel-trace --stats of the same command)", "",
        "| kernel | calls | avg us | min us | max us | % |", "|---|---|---|---|---|---|"]
for x in rows[:24]:
    out.append(f"| `{x['Name'][:110]}` | {x['Calls']} | {float(x['AverageNs']) / 1e3:.1f} | {float(x['MinNs']) / 1e3:.1f} | {float(x['MaxNs']) / 1e3:.1f} | {x['Percentage']} |")
out += ["", "Averages mix the overlapped steps with the few isolated steps bench.py runs for `roofline.isolated` (the minimum column).", "",
        "## Per launch: algorithmic bytes vs measured HBM-side traffic (PMC) and HIP-event duration (overlapped run)", "",
        "| bench label | calls/step | us | alg GB/s | TF | alg MB | PMC traffic MB | traffic/alg |", "|---|---|---|---|---|---|---|---|"]
for name, calls, ms, gbs, tf in ev:
    key = f"{name}|H{H}|L{L}|B{B}|{dtype}"
    alg = gbs * ms / calls * 1e-3 * 1e3  # MB per launch = GB/s * ms
    t = traffic.get(key, {}).get("traffic_bytes_per_launch")
    out.append(f"| {name} | {calls} | {1e3 * ms / calls:.1f} | {gbs:.0f} | {tf:.1f} | {alg:.1f} | {t / 1e6:.1f} | {t / 1e6 / alg:.2f} |" if t and alg > 0 else
               f"| {name} | {calls} | {1e3 * ms / calls:.1f} | {gbs:.0f} | {tf:.1f} | {alg:.1f} | - | - |")
open(os.path.join(root, "profiles", f"{tag}_summary.md"), "w").write("\n".join(out) + "\n")
print("wrote", f"profiles/{tag}_summary.md")
