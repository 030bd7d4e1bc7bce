"""GPU: csrc/device_prims.hip.h (the scans, the maximum and the stable radix sort of the preparation pass; hipCUB calls
until round 5) against numpy, through the p3r_test_* seam that only the knobs build of the library exports.  The
preparation parity suites (test_gpu_prep_device.py: device tables byte-equal to the host restatement's) run them inside
the product."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KNOBS = os.path.join(ROOT, "plonky3_recursion_amd", "knobs", "libp3r_hip.so")


@pytest.mark.gpu
def test_device_primitives_against_numpy():
    if not os.path.exists(KNOBS):
        pytest.skip("knobs build of the library is absent (__graft_entry__.build() makes it)")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "device_prims_cases.py")], capture_output=True, text=True,
                       env=dict(os.environ, P3R_LIB_PATH=KNOBS), timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "device_prims ok" in r.stdout


def test_product_library_does_not_export_the_seam():
    import ctypes
    lib = ctypes.CDLL(os.path.join(ROOT, "plonky3_recursion_amd", "libp3r_hip.so"))
    for name in ("p3r_test_exclusive_sum_u32", "p3r_test_exclusive_sum_u64", "p3r_test_reduce_max", "p3r_test_sort_pairs"):
        assert not hasattr(lib, name), name
