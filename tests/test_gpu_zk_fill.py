"""GPU: the tiled ZK fills (csrc/kernels_zk.hip.h::k_zk_fill_tiles - one ChaCha block per eight cells, parked in LDS and
written out column by column) against the per-cell kernels they replace (k_zk_randomize, k_zk_salts), cell for cell, on
matrices of one row to 2^15 rows, one to 300 columns, unit and non-unit strides, both fields
(tools/microbench/zk_fill_check.hip, built by __graft_entry__.build()).  The byte-equality of whole ZK proofs with the
oracle (tests/test_gpu_zk.py, tests/test_gpu_hiding_mmcs.py) covers the same kernels end to end."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "tools", "microbench", "zk_fill_check")


def test_tiled_fills_equal_the_per_cell_kernels():
    if not os.path.exists(EXE):
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-w", "-I",
                        os.path.join(ROOT, "plonky3_recursion_amd", "csrc"), "-o", EXE, EXE + ".hip"], check=True, timeout=600)
    r = subprocess.run([EXE], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "all shapes identical" in r.stdout and "MISMATCH" not in r.stdout
    assert r.stdout.count("0 of ") >= 38


def test_tiled_fills_with_cell_indices_beyond_2_to_the_32():
    """2^24 rows x 300 / 330 columns of one stream (5 x 10^9 cells, 40 GB for the two copies): the 64-bit index paths of
    the tiled kernel against the per-cell kernels (launched in two halves: a grid holds fewer than 2^32 threads), compared
    on the device."""
    r = subprocess.run([EXE, "--big"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert r.stdout.count(": 0 cells differ") == 4 and "all shapes identical" in r.stdout
