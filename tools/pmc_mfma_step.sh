#!/bin/bash
# Matrix-pipe occupancy of every kernel of the real training step by COUNTER (north star: "MFMA utilisation for attention / MLP vs gfx950 peak, evidenced
# by rocprof"): separate --pmc passes over bench.py (never combined with the trace domains gpurun refuses), then per kernel
#   mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 256 CUs x 4 SIMDs)      (the share of SIMD-cycles the matrix pipe worked)
# beside the launch count and average duration of the kernel-trace.  Writes gpurun_out/pmc_mfma/summary.txt; copy it to profiles/rNN_pmc_mfma_step.txt.
# usage (GPU box): bash tools/pmc_mfma_step.sh [extra bench.py arguments, e.g. --model vitl32 --fp8]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export MOFO_ROUTE_AB=0     # the step's default routes only (no in-process A/B of the round-5 routes)
mkdir -p gpurun_out/pmc_mfma
for pass in "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES" "GRBM_GUI_ACTIVE"; do
  n=$(echo $pass | cut -d' ' -f1)
  rocprofv3 --pmc $pass --kernel-trace --output-format csv -d gpurun_out/pmc_mfma/$n -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-encoder-step --no-kernel-events "$@" --no-calibration > gpurun_out/pmc_mfma/$n.log 2>&1 || echo "pass $n failed"
done
python3 - <<PY > gpurun_out/pmc_mfma/summary.txt
import csv, glob, collections, re
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(collections.Counter)
dur = collections.defaultdict(list)
for f in glob.glob("gpurun_out/pmc_mfma/*/*/*counter_collection.csv"):
    for row in csv.DictReader(open(f)):
        k = re.sub(r"\(anonymous namespace\)::", "", row["Kernel_Name"]); k = re.sub(r"\(.*", "", k).replace("void ", "")
        agg[k][row["Counter_Name"]] += float(row["Counter_Value"]); cnt[k][row["Counter_Name"]] += 1
for f in glob.glob("gpurun_out/pmc_mfma/GRBM_GUI_ACTIVE/*/*kernel_trace.csv"):
    for row in csv.DictReader(open(f)):
        k = re.sub(r"\(anonymous namespace\)::", "", row["Kernel_Name"]); k = re.sub(r"\(.*", "", k).replace("void ", "")
        dur[k].append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3)
rows = []
for k, d in agg.items():
    if "GRBM_GUI_ACTIVE" not in d or "SQ_VALU_MFMA_BUSY_CYCLES" not in d: continue
    gui = d["GRBM_GUI_ACTIVE"] / cnt[k]["GRBM_GUI_ACTIVE"] / 8.0
    busy = d["SQ_VALU_MFMA_BUSY_CYCLES"] / cnt[k]["SQ_VALU_MFMA_BUSY_CYCLES"]
    n = cnt[k]["SQ_VALU_MFMA_BUSY_CYCLES"]
    us = sum(dur[k]) / max(1, len(dur[k]))
    rows.append((n * us, k, n, us, busy / (gui * 1024.0) if gui else 0.0, d["SQ_INSTS_MFMA"] / cnt[k]["SQ_INSTS_MFMA"]))
print(f"{'kernel':74s} {'launches':>8s} {'avg us':>9s} {'mfma busy':>10s} {'MFMA insts / launch':>20s}")
for _, k, n, us, b, im in sorted(rows, reverse=True)[:40]:
    print(f"{k[:74]:74s} {n:8d} {us:9.1f} {b:10.1%} {im:20.0f}")
PY
cat gpurun_out/pmc_mfma/summary.txt
