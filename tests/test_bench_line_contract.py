"""The driver reads ONE JSON line from bench.py's stdout and keeps only an ~8 KB tail of it: the final line must be the
contract fields alone, under bench.CONTRACT_MAX_CHARS characters, whatever the detail dict has grown to (round 5's
20.8 KB line left BENCH_r05.json.parsed = null).  The reducer is a pure function, held to that here on the full result
dicts of earlier rounds (profiles/r05/bench_line_*.json are what bench.py measured then) and on a worst case."""
import copy
import glob
import json
import math
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

CANNED = sorted(glob.glob(os.path.join(ROOT, "profiles", "r05", "bench_line_*.json")))
REQUIRED = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
            "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "proof_verified", "proof_sha256"}


def _full(path):
    with open(path) as fh:
        txt = fh.read()
    return json.loads(txt.strip().splitlines()[-1]) if not txt.lstrip().startswith("{\n") else json.loads(txt)


def _no_nan(x):
    if isinstance(x, float):
        return math.isfinite(x)
    if isinstance(x, dict):
        return all(_no_nan(v) for v in x.values())
    if isinstance(x, list):
        return all(_no_nan(v) for v in x)
    return True


def test_canned_lines_exist():
    assert CANNED, "profiles/r05/bench_line_*.json are the canned inputs of this test"


@pytest.mark.parametrize("path", CANNED, ids=[os.path.basename(p) for p in CANNED])
def test_final_line_is_compact_and_complete(path):
    full = _full(path)
    if "proof_sha256" not in full:      # round-5 tree lines carry root_sha256 / root_verified
        full["proof_sha256"] = full.get("root_sha256")
        full["proof_verified"] = full.get("root_verified")
    text = bench.contract_line(full, "bench_detail.json")
    assert "\n" not in text
    assert len(text) < bench.CONTRACT_MAX_CHARS < 6001
    line = json.loads(text)
    assert REQUIRED <= set(line), REQUIRED - set(line)
    assert _no_nan(line)
    assert json.loads(json.dumps(line)) == line
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "data"):
        want = full[k]
        assert line[k] == pytest.approx(want, rel=1e-5) if isinstance(want, float) else line[k] == want
    assert line["config"]["workload"] and "model" not in line["config"]
    assert line["proof_sha256"] == full["proof_sha256"]
    if full.get("roofline"):
        r = line["roofline"]
        assert {"bound", "achieved", "peak", "unit", "frac", "traffic"} <= set(r)
        assert r["frac"] == pytest.approx(full["roofline"]["frac"], rel=1e-5)
        assert r["achieved"] / r["peak"] == pytest.approx(r["frac"], rel=1e-4)
    if full.get("cpu_baseline"):
        c = line["cpu_baseline"]
        assert {"value", "unit", "cores", "kind", "sample"} <= set(c)
        assert c["kind"] in ("port", "reference")


def test_headline_line_has_roofline_and_cpu_baseline():
    full = _full(os.path.join(ROOT, "profiles", "r05", "bench_line_final.json"))
    line = json.loads(bench.contract_line(full))
    assert line["roofline"]["kernel"] == "k_mmcs_hash_rows" and 0 < line["roofline"]["frac"] <= 1.0
    assert line["cpu_baseline"]["value"] > 0 and line["cpu_baseline"]["cores"] >= 1
    assert line["metric"] == json.load(open(os.path.join(ROOT, "BASELINE.json")))["metric"]
    assert line["vs_baseline"] is None


def test_oversized_prose_and_nan_are_contained():
    full = copy.deepcopy(_full(os.path.join(ROOT, "profiles", "r05", "bench_line_final.json")))
    full["config"]["workload"] = "w" * 5000
    full["cpu_baseline"]["sample"] = "s" * 5000
    full["roofline"]["traffic"] = float("nan")
    full["junk"] = {"k%d" % i: list(range(100)) for i in range(100)}
    text = bench.contract_line(full, "bench_detail.json")
    assert len(text) < bench.CONTRACT_MAX_CHARS
    line = json.loads(text)
    assert "junk" not in line and line["roofline"]["traffic"] is None


def test_detail_file_round_trip(tmp_path):
    full = _full(os.path.join(ROOT, "profiles", "r05", "bench_line_final.json"))
    p = tmp_path / "bench_detail.json"
    bench.write_detail(full, str(p))
    assert json.loads(p.read_text()) == full
