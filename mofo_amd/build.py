"""Build libmofo_hip.so (HIP kernels + C-ABI) in-tree for gfx950 with hipcc.  No JIT cache: the .so sits next to the
sources so that it travels with the repo snapshot to the GPU box."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libmofo_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
SOURCES = ["gemm.hip", "attention.hip", "layernorm.hip", "tokens.hip", "loss.hip", "optim.hip", "quant.hip", "capi.cpp", "comm.cpp"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=fast", "-Wall", "-Wno-unused-function",
         # keep MFMA accumulators in arch VGPRs: with the default AGPR form hipcc (ROCm 7.2) copied every accumulator in
         # and out of the AGPR file each loop iteration (100 v_accvgpr_* per 16 MFMAs in the GEMM main loop)
         "-mllvm", "-amdgpu-mfma-vgpr-form"]


# attention.hip: no SLP vectorisation.  hipcc packs neighbouring f32 multiplies of MFMA accumulator registers into v_pk_mul_f32
# plus v_mov shuffles; a packed f32 op issues in two passes anyway (guide MI355X_MICROARCH.md: an anti-lever beside MFMAs), and
# these kernels are bound by the SIMD's vector issue port.  (Inline-asm single multiplies are not an option: an asm VALU that
# reads an MFMA result gets none of the compiler's hazard wait states -- it read stale accumulators.)
EXTRA_FLAGS = {"attention.hip": ["-fno-slp-vectorize"]}


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    headers = [os.path.join(CSRC, "common.h"), os.path.join(CSRC, "gemm8.h"), os.path.join(CSRC, "gemm_k2.h"), os.path.join(CSRC, "gemm_r3.h"), os.path.join(CSRC, "gemm_r4.h"), os.path.join(HERE, "..", "include", "mofo_hip.h"), os.path.abspath(__file__)]
    objdir = os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    jobs = []
    for src in SOURCES:
        sp = os.path.join(CSRC, src)
        op = os.path.join(objdir, src + ".o")
        if force or _stale(op, [sp] + headers):
            cmd = [HIPCC] + FLAGS + EXTRA_FLAGS.get(src, []) + (["-x", "hip"] if src.endswith(".cpp") else []) + ["-c", sp, "-o", op]
            jobs.append(cmd)
    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed:\n" + " ".join(cmd) + "\n" + r.stdout + r.stderr)
        if verbose and r.stderr.strip():
            print(r.stderr)
    with ThreadPoolExecutor(max_workers=4) as ex:
        list(ex.map(run, jobs))
    objs = [os.path.join(objdir, s + ".o") for s in SOURCES]
    if force or jobs or _stale(LIB, objs):
        run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs + ["-ldl"])
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
