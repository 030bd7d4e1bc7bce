"""CPU: the AVX-512 host permutation of the transcript (csrc/host_poseidon2_simd.h) is bit-identical
to the scalar template, and the native verifier accepts the same proofs with either."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_avx512_permutation_equals_scalar(tmp_path):
    exe = str(tmp_path / "hp2")
    subprocess.run(["g++", "-O2", "-std=c++17", "-I", os.path.join(ROOT, "plonky3_recursion_amd", "csrc"),
                    os.path.join(ROOT, "tools", "microbench", "host_poseidon2_check.cpp"), "-o", exe], check=True)
    r = subprocess.run([exe], capture_output=True, text=True)
    if r.returncode == 77:
        pytest.skip(r.stdout.strip())
    assert r.returncode == 0, r.stdout + r.stderr
    assert "mismatches 0 of 20000" in r.stdout


def test_native_verifier_with_scalar_host_permutation():
    """Accepting and rejecting cases of tests/test_native_verifier.py with P3R_HOST_SIMD=0."""
    env = dict(os.environ, P3R_HOST_SIMD="0")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_native_verifier.py"), "-q", "-x",
                        "-k", "wire_round_trip or unsatisfied", "-p", "no:cacheprovider"], env=env, cwd=ROOT,
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "passed" in r.stdout
