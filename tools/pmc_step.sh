#!/bin/bash
# HBM traffic counters of the kernels of the real training step (bench.py), separate --pmc passes (FETCH_SIZE and WRITE_SIZE do
# not fit one pass; --pmc is never combined with the trace domains gpurun refuses).  Writes gpurun_out/pmc_step/summary.json:
#   per kernel name and per bench.py kernel CLASS: launches, fetched / written MB per launch (FETCH_SIZE x2: gfx950 counts wide
#   coalesced reads at one half, guide MI355X_MICROARCH.md "HBM"), and _meta.csrc_sha256 = the sha of the kernel sources the
#   profile belongs to (bench.py reports `traffic` only while that sha matches).  Copy it to profiles/rNN_pmc_step_traffic.json.
# usage (GPU box): bash tools/pmc_step.sh
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/pmc_step
export MOFO_ROUTE_AB=0     # the step's default routes only (no in-process A/B of the round-5 routes)
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/pmc_step/$c -- python3 bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-encoder-step --no-kernel-events --no-calibration > gpurun_out/pmc_step/$c.log 2>&1
done
python3 - <<PY
import csv, glob, collections, json, re, sys
sys.path.insert(0, ".")
import bench
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(lambda: collections.Counter())
for f in glob.glob("gpurun_out/pmc_step/*/*/*counter_collection.csv"):
    for row in csv.DictReader(open(f)):
        k = re.sub(r"\(anonymous namespace\)::", "", row["Kernel_Name"]); k = re.sub(r"\(.*", "", k).replace("void ", "")
        agg[k][row["Counter_Name"]] += float(row["Counter_Value"]); cnt[k][row["Counter_Name"]] += 1
out = {}
for k, d in agg.items():
    n = max(cnt[k].values())
    # FETCH_SIZE / WRITE_SIZE are in KB; FETCH_SIZE counts wide coalesced reads at 1/2 on gfx950 (guide) -> x2
    out[k] = {"launches": n, "fetch_MB_per_launch": round(2 * d.get("FETCH_SIZE", 0) / max(1, cnt[k]["FETCH_SIZE"]) / 1e3, 2),
              "write_MB_per_launch": round(d.get("WRITE_SIZE", 0) / max(1, cnt[k]["WRITE_SIZE"]) / 1e3, 2)}
# bench.py's kernel classes = one C-ABI entry each; a class may run several kernel variants (GEMM tilings): launch-weighted mean.
# GEMM templates start <LA, LB, EPI, ...>: LA/LB 0 = ROW, 1 = COL operand.
# (round 6: a class launch = one C-ABI call; mofo_gemm_wgrad_sliced is TWO kernels, gemm_r4 + the slab reduce: the reduce kernel's bytes
# count, its launches do not -- `riders`)
riders = r"^wgrad_slab_reduce_kernel"
classes = {"gemm_tn_wgrad_f32": r"^gemm\w*_kernel<1, 1, 5|^gemm_r4_kernel|^wgrad_slab_reduce_kernel", "gemm_nn_bf16": r"^gemm\w*_kernel<0, 1, 0", "gemm_nn_dgelu": r"^gemm\w*_kernel<0, 1, 4",
           "gemm_nt_bf16": r"^gemm\w*_kernel<0, 0, 0", "gemm_nt_bias_gelu": r"^gemm\w*_kernel<0, 0, 1", "gemm_nt_resid_f32": r"^gemm\w*_kernel<0, 0, 2",
           "gemm_nt_resid_bf16": r"^gemm\w*_kernel<0, 0, 6", "attn_fwd": r"^attn_q_kernel<\d, 0", "attn_bwd_dq": r"^attn_q_kernel<\d, 1",
           "attn_bwd_dkv": r"^attn_dkv_kernel", "attn_bwd": r"^attn_bwd_fused_kernel",
           "ln_fwd": r"^ln_fwd_kernel", "ln_bwd": r"^ln_bwd_kernel", "adamw": r"^adamw_kernel", "target_mse": r"^target_mse_kernel"}
cls = {}
for name, rx in classes.items():
    ks = [k for k in out if re.search(rx, k)]
    n = sum(out[k]["launches"] for k in ks if not re.search(riders, k))
    if n:
        cls[name] = {"launches": n, "kernels": ks,
                     "fetch_MB_per_launch": round(sum(out[k]["fetch_MB_per_launch"] * out[k]["launches"] for k in ks) / n, 2),
                     "write_MB_per_launch": round(sum(out[k]["write_MB_per_launch"] * out[k]["launches"] for k in ks) / n, 2)}
for k, v in sorted(out.items(), key=lambda kv: -(kv[1]["fetch_MB_per_launch"] + kv[1]["write_MB_per_launch"]) * kv[1]["launches"])[:40]:
    print(f"{k[:70]:70s} {v}")
json.dump({"_meta": {"csrc_sha256": bench.csrc_sha(), "command": "MOFO_ROUTE_AB=0 python3 bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-encoder-step --no-kernel-events --no-calibration",
                     "note": "MB per launch; fetch = FETCH_SIZE x 2 (gfx950 correction), write = WRITE_SIZE; separate rocprofv3 --pmc passes"},
           "classes": cls, "kernels": out}, open("gpurun_out/pmc_step/summary.json", "w"), indent=1)
PY
