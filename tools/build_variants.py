#!/usr/bin/env python3
"""Build side-by-side variants of libmofo_hip.so for same-process A/B (tools/attn_ab.py --libs ..., MOFO_HIP_LIB=...).
usage: build_variants.py <source.hip> name1[:-DFLAG[,-DFLAG2...]] name2[:...] ...
Only <source.hip> is recompiled per variant (with the project's flags + the variant's); the other objects come from the normal
in-tree build (mofo_amd/build/).  Output: tools/_ab/<name>.so (git-ignored, travels to the GPU box).  --asm also keeps
/tmp/abbuild/<name>.s (device assembly) for reading."""
import os, subprocess, sys
from concurrent.futures import ThreadPoolExecutor
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mofo_amd import build as b

args = [a for a in sys.argv[1:] if a != "--asm"]
keep_asm = "--asm" in sys.argv
srcname, specs = args[0], args[1:]
b.build()
csrc, objdir, out = os.path.join(ROOT, "mofo_amd", "csrc"), os.path.join(ROOT, "mofo_amd", "build"), os.path.join(ROOT, "tools", "_ab")
os.makedirs(out, exist_ok=True), os.makedirs("/tmp/abbuild", exist_ok=True)


def mk(spec):
    name, _, fl = spec.partition(":")
    extra = [f for f in fl.split(",") if f]
    base = [b.HIPCC] + b.FLAGS + b.EXTRA_FLAGS.get(srcname, []) + extra
    obj = f"/tmp/abbuild/{name}_{srcname}.o"
    r = subprocess.run(base + ["-c", os.path.join(csrc, srcname), "-o", obj], capture_output=True, text=True)
    if r.returncode:
        return f"{name}: compile FAILED\n{r.stderr[-3000:]}"
    if keep_asm:
        subprocess.run(base + ["-S", "--cuda-device-only", os.path.join(csrc, srcname), "-o", f"/tmp/abbuild/{name}.s"], capture_output=True, text=True)
    objs = [obj if s == srcname else os.path.join(objdir, s + ".o") for s in b.SOURCES]
    r = subprocess.run([b.HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", os.path.join(out, name + ".so")] + objs + ["-ldl"], capture_output=True, text=True)
    return f"{name}: {'ok' if r.returncode == 0 else 'link FAILED ' + r.stderr[-500:]}  flags {extra}"


with ThreadPoolExecutor(4) as ex:
    for line in ex.map(mk, specs):
        print(line)
