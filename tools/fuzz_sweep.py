"""One-off wide sweep of tests/test_gpu_circuit_fuzz.py's check (random circuits: device runner and
host preprocessing against the oracle) over many more seeds than the test suite runs.

usage: python tools/fuzz_sweep.py [first_seed] [count] [n_ops]
"""
import sys

sys.path.insert(0, "tests")
sys.path.insert(0, ".")
import oracle_lib
import test_gpu_circuit_fuzz as t

first = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
count = int(sys.argv[2]) if len(sys.argv) > 2 else 200
n_ops = int(sys.argv[3]) if len(sys.argv) > 3 else 400
oracle = oracle_lib.Oracle()
fn = getattr(t.test_random_circuits_run_and_preprocess_like_the_oracle, "__wrapped__",
             t.test_random_circuits_run_and_preprocess_like_the_oracle)
for field, lo in (("koala-bear", first), ("baby-bear", first + count)):
    fn(oracle, field, range(lo, lo + count), n_ops)
    print(f"{field}: seeds {lo}..{lo + count - 1} x {n_ops} ops agree with the oracle", flush=True)
