#!/bin/bash
# Counters of gemm_r4_kernel on ONE group shape (tools/wgrad_dec_ab.py --quick --only-enc | --only-dec): L2 hits / misses, fabric bytes,
# texture addresser / L1 stalls, matrix pipe, wait states.  Separate --pmc passes, never combined with trace domains other than kernel-trace.
# usage (GPU box): bash tools/pmc_r4.sh enc|dec
which=$1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/pmc_r4_$which; rm -rf $out; mkdir -p $out
for pass in "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum" "FETCH_SIZE" "WRITE_SIZE" \
            "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM_RD SQ_WAIT_INST_LDS" \
            "TA_TA_BUSY_sum TD_TD_BUSY_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum" \
            "TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TD_TCP_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum" \
            "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TA_TCP_STATE_READ_sum TCP_TCC_READ_REQ_LATENCY_sum" \
            "GRBM_GUI_ACTIVE"; do
  n=$(echo $pass | cut -d' ' -f1)
  rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $out/$n -- python3 tools/wgrad_dec_ab.py 1 --quick --only-$which > $out/$n.log 2>&1 || echo "pass $n failed (see $out/$n.log)"
done
python3 - $out <<'PY'
import csv, glob, collections, sys
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(collections.Counter)
for f in glob.glob(out + "/*/*/*counter_collection.csv"):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        if "gemm_r4" not in k and "slab_reduce" not in k: continue
        k = k.replace("(anonymous namespace)::", "").replace("void ", "")[:48]
        agg[k][row["Counter_Name"]] += float(row["Counter_Value"]); cnt[k][row["Counter_Name"]] += 1
for k, d in agg.items():
    v = {c: d[c] / cnt[k][c] for c in d}
    gui = v.get("GRBM_GUI_ACTIVE", 0) / 8
    print("==", k, f"launches {max(cnt[k].values())}, {gui:.0f} clk per launch")
    for c in sorted(v):
        per = f"   {v[c] / gui / 256:10.3f} per CU-cycle" if gui else ""
        print(f"   {c:40s} {v[c]:16.0f}{per}")
    if "TCC_HIT_sum" in v:
        print(f"   L2 hit rate {v['TCC_HIT_sum'] / max(1.0, v['TCC_HIT_sum'] + v['TCC_MISS_sum']):.3f}")
PY
