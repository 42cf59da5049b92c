#!/bin/bash
# Issue-side counters for one kernel shape: how busy are the SIMD's vector issue port and the MFMA pipe, and do they overlap?
# usage (GPU box): bash tools/pmc_issue.sh <tag> <args to one_gemm.py>       e.g.  pmc_issue.sh attnb attnb 32 1568 6
# Prints per kernel and launch: MFMA-busy, VALU-active, co-execution cycles as fractions of the SIMD cycles of the launch.
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/pmci_$tag
for pass in "SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VALU2 SQ_ACTIVE_INST_MISC SQ_BUSY_CU_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA" \
            "SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32 SQ_THREAD_CYCLES_VALU SQ_INST_LEVEL_LDS" \
            "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_BUSY_CYCLES" \
            "GRBM_GUI_ACTIVE"; do
  n=$(echo $pass | cut -d' ' -f1)
  rocprofv3 --pmc $pass --kernel-trace --output-format csv -d gpurun_out/pmci_$tag/$n -- python3 tools/one_gemm.py "$@" > gpurun_out/pmci_$tag/$n.log 2>&1
done
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(collections.Counter)
for f in glob.glob("gpurun_out/pmci_$tag/*/*/*counter_collection.csv"):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        if "gemm" not in k and "attn" not in k: continue
        k = k.replace("(anonymous namespace)::", "").replace("void ", "")[:48]
        agg[k][row["Counter_Name"]] += float(row["Counter_Value"]); cnt[k][row["Counter_Name"]] += 1
for k, d in agg.items():
    v = {c: d[c] / cnt[k][c] for c in d}
    simd_cyc = v.get("GRBM_GUI_ACTIVE", 0) / 8 * 1024          # GRBM counts per XCD; 1024 SIMDs
    print("==", k, f"launches {max(cnt[k].values())}")
    for c in sorted(v):
        print(f"   {c:30s} {v[c]:16.0f}")
    if simd_cyc:
        q = lambda c, mul=1.0: v.get(c, 0) * mul / simd_cyc
        print(f"   -> per SIMD cycle: MFMA busy {q('SQ_VALU_MFMA_BUSY_CYCLES'):.3f}  coexec {q('SQ_VALU_MFMA_COEXEC_CYCLES'):.3f}  "
              f"ACTIVE_INST_VALU x4 {q('SQ_ACTIVE_INST_VALU', 4):.3f}  VALU2 x4 {q('SQ_ACTIVE_INST_VALU2', 4):.3f}  LDS x4 {q('SQ_ACTIVE_INST_LDS', 4):.3f}  "
              f"wave-cycles x4 {q('SQ_WAVE_CYCLES', 4):.2f} waves/SIMD  wait-any x4 {q('SQ_WAIT_ANY', 4):.2f}")
        if v.get("SQ_INSTS_MFMA"):
            print(f"   -> VALU per MFMA {v['SQ_INSTS_VALU'] / v['SQ_INSTS_MFMA']:.2f} (trans {v.get('SQ_INSTS_VALU_TRANS_F32', 0) / v['SQ_INSTS_MFMA']:.2f}, fma {v.get('SQ_INSTS_VALU_FMA_F32', 0) / v['SQ_INSTS_MFMA']:.2f}, "
                  f"add {v.get('SQ_INSTS_VALU_ADD_F32', 0) / v['SQ_INSTS_MFMA']:.2f}, mul {v.get('SQ_INSTS_VALU_MUL_F32', 0) / v['SQ_INSTS_MFMA']:.2f}, cvt {v.get('SQ_INSTS_VALU_CVT', 0) / v['SQ_INSTS_MFMA']:.2f}, int {v.get('SQ_INSTS_VALU_INT32', 0) / v['SQ_INSTS_MFMA']:.2f})")
PY
