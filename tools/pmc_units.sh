#!/bin/bash
# Which unit is busy?  LDS array, texture addresser / data (TA / TD), L1 (TCP) stalls, MFMA pipe, wait states -- one kernel shape.
# usage (GPU box): bash tools/pmc_units.sh <tag> <args to one_gemm.py>      e.g.  pmc_units.sh decqkv nt 50176 1152 384
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/pmcu_$tag
for pass in "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" \
            "SQ_INSTS_LDS_LOAD SQ_INSTS_LDS_STORE SQ_INSTS_LDS_LOAD_BANDWIDTH SQ_INSTS_LDS_STORE_BANDWIDTH SQ_INST_LEVEL_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_BUSY_CU_CYCLES" \
            "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR" \
            "TA_TA_BUSY_sum TD_TD_BUSY_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum" \
            "TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TD_TCP_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum" \
            "TCP_LFIFO_STALL_CYCLES_sum TCP_RFIFO_STALL_CYCLES_sum TCP_TCP_TA_ADDR_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum" \
            "GRBM_GUI_ACTIVE"; do
  n=$(echo $pass | cut -d' ' -f1)
  rocprofv3 --pmc $pass --kernel-trace --output-format csv -d gpurun_out/pmcu_$tag/$n -- python3 tools/one_gemm.py "$@" > gpurun_out/pmcu_$tag/$n.log 2>&1 || echo "pass $n failed (see gpurun_out/pmcu_$tag/$n.log)"
done
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(collections.Counter)
for f in glob.glob("gpurun_out/pmcu_$tag/*/*/*counter_collection.csv"):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        if "gemm" not in k and "attn" not in k: continue
        k = k.replace("(anonymous namespace)::", "").replace("void ", "")[:48]
        agg[k][row["Counter_Name"]] += float(row["Counter_Value"]); cnt[k][row["Counter_Name"]] += 1
for k, d in agg.items():
    v = {c: d[c] / cnt[k][c] for c in d}
    gui = v.get("GRBM_GUI_ACTIVE", 0) / 8          # per-XCD cycles of the launch
    print("==", k, f"launches {max(cnt[k].values())}, {gui:.0f} clk per launch")
    for c in sorted(v):
        per = ""
        if gui:
            per = f"   {v[c] / gui / 256:8.3f} per CU-cycle   {v[c] / gui / 1024:8.3f} per SIMD-cycle"
        print(f"   {c:40s} {v[c]:16.0f}{per}")
PY
