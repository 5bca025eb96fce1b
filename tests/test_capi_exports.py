"""CPU-side checks of the C-ABI library: it loads, exports every symbol include/mpc_batch.h
declares, refuses to create a handle without a GPU (no silent CPU fallback), and the product
package never touches the oracle."""
import ctypes
import os
import re

import pytest

from mpc_amd import capi, variants

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "mpc_batch.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mpc_(?:batch|spec)_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    L = capi.lib()
    names = _declared_symbols()
    assert len(names) >= 13
    for n in names:
        assert hasattr(L, n), f"libmpc_amd.so does not export {n}"
    assert set(names) == set(capi.EXPORTS)
    assert b"gfx950" in L.mpc_batch_version()


def test_spec_variants_match_python_table():
    for name in variants.VARIANTS:
        c = capi.spec_variant(name)
        p = variants.spec(name)
        assert bytes(c) == bytes(p), name
    with pytest.raises(capi.MpcError):
        capi.spec_variant("mobile_robot_99")


def test_no_cpu_fallback_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(capi.MpcError):
        capi.BatchSolver(variants.spec("mobile_robot"), 4)


def test_null_handle_calls_are_rejected():
    L = capi.lib()
    assert L.mpc_batch_solve(None, 1, None, None, None, None, None, None, None, None) == -1
    assert L.mpc_batch_reset(None, None) == -1
    assert L.mpc_batch_horizon(None) == -1
    L.mpc_batch_free(None)
    h = ctypes.c_void_p()
    assert L.mpc_batch_create(ctypes.byref(h), None, 4, None, 0) == -1 and not h


def test_product_package_does_not_reference_the_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "mpc_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h", "Makefile")):
                src = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "oracle/" not in src and "oracle_lib" not in src and "mpc_oracle" not in src, f
