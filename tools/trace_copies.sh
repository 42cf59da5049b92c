#!/bin/bash
# which memory copies and HIP runtime calls does one training step make?  (rocprofv3 memory-copy + HIP runtime trace over bench.py, no counters)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export MOFO_ROUTE_AB=0
rm -rf gpurun_out/mc; mkdir -p gpurun_out/mc
rocprofv3 --memory-copy-trace --hip-runtime-trace --stats --output-format csv -d gpurun_out/mc -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-encoder-step --no-kernel-events --no-calibration > gpurun_out/mc/bench.json 2> gpurun_out/mc/bench.err
ls gpurun_out/mc/*/
python3 - <<'PY'
import csv, glob, collections
for f in glob.glob("gpurun_out/mc/*/*memory_copy_trace.csv"):
    c = collections.Counter()
    rows = list(csv.DictReader(open(f)))
    print(f, len(rows), rows[0].keys() if rows else None)
    for r in rows:
        c[(r.get("Direction"), r.get("Bytes") or r.get("Size"))] += 1
    for k, v in c.most_common(15): print(k, v)
for f in glob.glob("gpurun_out/mc/*/*stats.csv"):
    print("==", f)
    for r in list(csv.DictReader(open(f)))[:12]: print({k: r[k] for k in list(r)[:6]})
PY
