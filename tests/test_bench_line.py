"""The one line bench.py prints must stay small enough for the driver to keep it whole (round 3: 21.9 KB, of which the driver kept the last 8 KB and
parsed nothing) and must carry the measurement contract's fields."""
import importlib.util
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("lum_bench", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_headline_of_a_full_record_is_small_and_complete():
    b = _bench()
    with open(os.path.join(ROOT, "profiles", "r03_bench.json")) as f:  # a full record of a real run (22 KB)
        detail = json.load(f)
    detail["value_exact"] = {"value": 2612.3456789, "unit": "Mrays/s", "flavour": "exact", "steps": 4, "warmup": 1, "ms_per_step": 471.23456789, "samples_per_s": 140812345.678}
    line = json.dumps(b.headline(detail))
    assert len(line) <= 4096
    h = json.loads(line)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in h, k
    assert h["config"]["workload"].startswith("C3")
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "avg_launch_ms"):
        assert k in h["roofline"], k
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in h["cpu_baseline"], k
    assert h["value_exact"]["flavour"] == "exact"
    assert set(h["secondary"]) == {"example", "scan"}
    assert abs(h["value"] - detail["value"]) / detail["value"] < 1e-4  # rounding to 5 digits only


def test_shade_is_priced_against_the_guides_vector_peak():
    b = _bench()
    assert b.VALU_PEAK_GINST == 1228.8 and b.VALU_MEASURED_FMA_GINST == 614.4
