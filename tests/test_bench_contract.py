"""The bench line's contract, checked on the committed output of the default command (profiles/): the keys the driver and the
judge read, their types, and the internal consistency of the roofline figures."""
import glob
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _latest():
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r01_*_bench_default*.json")))
    assert files, "no committed bench line under profiles/"
    return json.load(open(files[-1]))


def test_bench_line_has_the_contract_keys():
    d = _latest()
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["unit"] == "episodes/s" and d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["data"] == "synthetic" and "workload" in d["config"] and "model" not in d["config"]
    assert "20x20" in d["config"]["workload"] and "32 items" in d["config"]["workload"] and "400" in d["config"]["workload"]
    assert d["n_gpus"] == 1 and d["value"] > 0 and d["ms_per_step"] > 0


def test_roofline_object_is_consistent():
    r = _latest()["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s")
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    # achieved = algorithmic flops per launch / mean launch duration
    assert abs(r["achieved"] - r["flops_per_launch"] / (r["ms_per_launch"] * 1e-3) / 1e12) < 1e-6 * r["achieved"]
    assert r["traffic"] is None or r["traffic"] >= 0.9 * r["algorithmic_bytes_per_launch"]


def test_cpu_baseline_object():
    c = _latest()["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] in ("reference", "port") and c["cores"] >= 1 and c["value"] > 0 and c["unit"] == "episodes/s"
