"""CPU-side checks of the drop-in boundary: the shared library loads and exports every symbol include/mofo_hip.h
declares, the ctypes table covers exactly that set, and the product path refuses to run without a GPU
(no silent CPU fallback)."""
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "mofo_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(mofo_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_exported():
    from mofo_amd import _lib
    lib = _lib.load()
    names = _declared()
    assert len(names) >= 16
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/mofo_hip.h but not exported"
    assert sorted(_lib.EXPORTS) == names, "ctypes table and header disagree"
    assert lib.mofo_version() == 1


def test_library_is_in_tree_and_hip_only():
    from mofo_amd import _lib
    assert os.path.dirname(_lib.LIB_PATH) == os.path.join(ROOT, "mofo_amd")
    out = os.popen(f"/opt/rocm/lib/llvm/bin/llvm-readelf -d {_lib.LIB_PATH} 2>/dev/null || readelf -d {_lib.LIB_PATH}").read()
    needed = " ".join(l for l in out.splitlines() if "NEEDED" in l)       # (addresses elsewhere in the dump may spell "c10")
    assert "libamdhip64" in needed
    assert "torch" not in needed and "c10" not in needed      # C-ABI: no torch types behind the boundary


def test_product_path_never_imports_oracle():
    pkg = os.path.join(ROOT, "mofo_amd")
    for fn in os.listdir(pkg):
        if fn.endswith(".py"):
            txt = open(os.path.join(pkg, fn)).read()
            assert not re.search(r"^\s*(from|import)\s+oracle\b", txt, flags=re.M), fn


def test_no_cpu_fallback():
    from mofo_amd import ops
    a = torch.zeros(64, 64, dtype=torch.bfloat16)
    with pytest.raises(ValueError, match="GPU"):
        ops.gemm(ops.GEMM_NT, ops.EPI_BF16, a, a, a.clone())
