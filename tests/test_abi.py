"""CPU: the C-ABI library loads and exports exactly the symbols include/p3r.h declares
(no compute calls: there is no GPU here)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    src = open(os.path.join(ROOT, "include/p3r.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(p3r_[a-z0-9_]+)\s*\(", src)))


def test_header_declares_entry_points():
    syms = header_symbols()
    for must in ("p3r_create", "p3r_destroy", "p3r_last_error", "p3r_poseidon2_permute_batch",
                 "p3r_poseidon2_trace_fill", "p3r_mmcs_commit", "p3r_coset_lde"):
        assert must in syms


def test_library_exports_every_declared_symbol():
    from plonky3_recursion_amd import _lib
    assert os.path.exists(_lib.LIB_PATH), "run __graft_entry__.build() first"
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for s in header_symbols():
        assert hasattr(lib, s), f"{s} declared in include/p3r.h but not exported"


def test_binding_table_matches_header():
    from plonky3_recursion_amd import _lib
    assert sorted(_lib.SIGNATURES) == header_symbols()


def test_create_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import plonky3_recursion_amd as p3r
    with pytest.raises(p3r.P3rError) as e:
        p3r.Context()
    assert "no CPU fallback" in str(e.value) or "HIP" in str(e.value)
