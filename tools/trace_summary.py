#!/usr/bin/env python3
"""aggregate a rocprofv3 kernel_trace.csv by (kernel, grid): launches/step, avg us, ms/step.  usage: trace_summary.py <csv> <steps>"""
import csv, sys, collections, re
rows = list(csv.DictReader(open(sys.argv[1])))
steps = int(sys.argv[2])
agg = collections.defaultdict(lambda: [0, 0.0])
for r in rows:
    name = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
    name = re.sub(r"\(.*", "", name)[:60]
    g = (r.get("Grid_Size_X") or r.get("Grid_Size") or "?")
    wg = r.get("Workgroup_Size_X") or r.get("Workgroup_Size") or "?"
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    a = agg[(name, g, wg)]
    a[0] += 1; a[1] += d
tot = sum(a[1] for a in agg.values())
print(f"total kernel time {tot/steps/1e3:.3f} ms/step over {steps} steps")
for (name, g, wg), (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:45]:
    print(f"{name:60s} grid={g:>9s} wg={wg:>4s} n/step={n/steps:6.1f} avg={t/n:8.1f} us  {t/steps/1e3:7.3f} ms/step")
