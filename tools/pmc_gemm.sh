#!/bin/bash
# counters for one kernel shape; separate --pmc passes (8 SQ slots per pass); usage: pmc_gemm.sh <tag> <args to one_gemm.py>
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/pmc_$tag
for pass in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA" \
            "SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_LDS" \
            "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_UNALIGNED_STALL SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_LDS_DATA_FIFO_FULL" \
            "GRBM_GUI_ACTIVE GRBM_TA_BUSY FETCH_SIZE" "WRITE_SIZE GRBM_TC_BUSY"; do
  n=$(echo $pass | cut -d' ' -f1)
  rocprofv3 --pmc $pass --kernel-trace --output-format csv -d gpurun_out/pmc_$tag/$n -- python3 tools/one_gemm.py "$@" > gpurun_out/pmc_$tag/$n.log 2>&1
done
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for f in glob.glob("gpurun_out/pmc_$tag/*/*/*counter_collection.csv"):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        if "gemm_kernel" not in k and "attn" not in k: continue
        k = k[:70]
        agg[k][row["Counter_Name"]] += float(row["Counter_Value"])
        if row["Counter_Name"] in ("SQ_WAVES", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_LDS_BANK_CONFLICT", "GRBM_GUI_ACTIVE", "WRITE_SIZE"): cnt[(k, row["Counter_Name"])] += 1
for k, d in agg.items():
    print("==", k)
    for c, v in sorted(d.items()):
        n = max(1, max(cnt[(k, x)] for x in ("SQ_WAVES", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_LDS_BANK_CONFLICT", "GRBM_GUI_ACTIVE", "WRITE_SIZE")))
        print(f"   {c:32s} {v/ n:16.1f} per launch")
PY
