#!/bin/bash
# list the non-libmofo kernels / copies of the timed steps (torch fills, copies) with their grid sizes: rocprofv3 kernel trace
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/misc_trace
rm -rf $OUT; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $OUT -- python3 bench.py --no-cpu-baseline --no-encoder-step --no-kernel-events --steps 4 --warmup 2 > $OUT/bench.log 2>&1
python3 - <<'PY'
import csv, glob, os
out = os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out/misc_trace"
kt = sorted(glob.glob(out + "/**/*kernel_trace.csv", recursive=True), key=os.path.getmtime)[-1]
rows = list(csv.DictReader(open(kt)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# find adamw launches = step boundaries
idx = [i for i, r in enumerate(rows) if "adamw_kernel" in r["Kernel_Name"]]
print("steps found", len(idx))
a, b = idx[-2], idx[-1]
t0 = int(rows[a]["End_Timestamp"])
with open(out + "/last_step.txt", "w") as f:
    prev_end = t0
    for r in rows[a + 1:b + 1]:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        name = r["Kernel_Name"][:70]
        f.write(f"{(s - t0) / 1e3:10.1f} us  dur {(e - s) / 1e3:8.1f}  gap {(s - prev_end) / 1e3:7.1f}  grid {r.get('Grid_Size_X', r.get('Grid_Size', '?')):>9}  {name}\n")
        prev_end = max(prev_end, e)
mc = sorted(glob.glob(out + "/**/*memory_copy_trace.csv", recursive=True), key=os.path.getmtime)
if mc:
    m = list(csv.DictReader(open(mc[-1])))
    t1 = int(rows[b]["End_Timestamp"])
    with open(out + "/last_step_copies.txt", "w") as f:
        for r in m:
            s = int(r["Start_Timestamp"])
            if t0 <= s <= t1 + 200000:
                f.write(f"{(s - t0) / 1e3:10.1f} us dur {(int(r['End_Timestamp']) - s) / 1e3:7.1f} {r.get('Direction', '')} bytes {r.get('Bytes', r.get('Size', '?'))}\n")
PY
