#!/bin/bash
# rocprofv3 kernel-trace stats of the real training step (bench.py, nothing instrumented in the process) under two sets of routing
# switches, same box, back to back: per kernel name the launches per step, average duration and ms per step.
# usage (GPU box): bash tools/trace_routes.sh NAME_A "ENV_A" NAME_B "ENV_B"   (ENV = "K=V K=V"; write "-" for none)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export MOFO_ROUTE_AB=0
STEPS=12; WARM=4
run() {
  name=$1; envs=$2
  rm -rf gpurun_out/trace_$name; mkdir -p gpurun_out/trace_$name
  if [ "$envs" != "-" ]; then export $envs; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/trace_$name -- python3 bench.py --steps $STEPS --warmup $WARM --no-cpu-baseline --no-encoder-step --no-kernel-events --no-calibration > gpurun_out/trace_$name/bench.json 2> gpurun_out/trace_$name/bench.err
  if [ "$envs" != "-" ]; then for kv in $envs; do unset ${kv%%=*}; done; fi
}
run "$1" "$2" && run "$3" "$4"
python3 - "$1" "$3" $STEPS $WARM <<'PY'
import csv, glob, sys, re, json, collections
a, b, steps, warm = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
def load(name):
    f = glob.glob(f"gpurun_out/trace_{name}/*/*kernel_stats.csv")
    d = {}
    for row in csv.DictReader(open(f[0])):
        k = re.sub(r"\(anonymous namespace\)::", "", row["Name"]); k = re.sub(r"\(.*", "", k).replace("void ", "")[:70]
        d[k] = (int(row["Calls"]), float(row["TotalDurationNs"]))
    line = open(f"gpurun_out/trace_{name}/bench.json").read().strip().splitlines()[-1]
    return d, json.loads(line)
da, ja = load(a); db, jb = load(b)
n = steps + warm + 0.0
print(f"# tools/trace_routes.sh: rocprofv3 --kernel-trace --stats over bench.py ({steps} timed + {warm} warm-up steps; per-step figures divide by all steps run)")
print(f"# {a}: {ja['ms_per_step']} ms per step, {ja['value']} clips/s | {b}: {jb['ms_per_step']} ms per step, {jb['value']} clips/s")
print(f"{'kernel':72s} {'calls/step':>10s} {a+' avg us':>14s} {a+' ms/step':>14s} | {'calls/step':>10s} {b+' avg us':>14s} {b+' ms/step':>14s}")
keys = sorted(set(da) | set(db), key=lambda k: -(da.get(k, (0, 0))[1] + db.get(k, (0, 0))[1]))
ta = tb = 0.0
for k in keys:
    ca, na = da.get(k, (0, 0.0)); cb, nb = db.get(k, (0, 0.0))
    ta += na; tb += nb
    if max(na, nb) / n < 2e3: continue
    print(f"{k:72s} {ca / n:10.1f} {na / max(ca, 1) / 1e3:14.1f} {na / n / 1e6:14.3f} | {cb / n:10.1f} {nb / max(cb, 1) / 1e3:14.1f} {nb / n / 1e6:14.3f}")
print(f"{'sum of all kernels (ms per step, incl. set-up kernels of the first steps)':72s} {'':10s} {'':14s} {ta / n / 1e6:14.3f} | {'':10s} {'':14s} {tb / n / 1e6:14.3f}")
PY
