"""bench.py's emitter: the ONE JSON line must stay small enough for the driver (it keeps an 8 KB tail of stdout and parses the
line out of it - round 4's 38 KB line came back `parsed: null`) and must be strict JSON; every sentence belongs to the sidecar."""
import json
import os

import bench

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _canned():
    """round 4's full record (seven workloads, paragraphs of prose in it) - the input that broke the driver's parser"""
    return json.load(open(os.path.join(ROOT, "profiles", "r04_bench_full_k20.json")))


def test_line_is_small_strict_json_with_the_contract_keys():
    full = _canned()
    assert len(json.dumps(full)) > 30000
    line = bench.compact_line(bench._finite(full))
    assert "\n" not in line and len(line) < bench.LINE_LIMIT < 8000
    j = json.loads(line, parse_constant=lambda c: (_ for _ in ()).throw(ValueError(c)))  # NaN / Infinity are not JSON
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
              "roofline", "cpu_baseline"):
        assert k in j, k
    assert j["value"] == float(f"{full['value']:.6g}") and j["ms_per_step"] == float(f"{full['ms_per_step']:.6g}")
    assert j["config"]["workload"] == full["config"]["workload"] and "model" not in j["config"]
    rf = j["roofline"]
    assert rf["bound"] == "hbm" and rf["unit"] == "GB/s" and rf["peak"] == 8000.0
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-5 and 0 < rf["frac"] <= 1.0
    assert rf["kernel_ms"] <= j["ms_per_step"]
    cb = j["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0 and cb["parity_on_sample"] is True
    # the secondaries are numbers (and a unit), never prose
    assert set(j["secondary"]) == set(full["secondary"])
    for nm, sec in j["secondary"].items():
        for k, v in sec.items():
            assert isinstance(v, (int, float, bool)) or k == "unit", (nm, k, v)
        assert sec["value"] > 0 and sec["kernel_ms"] <= sec["ms_per_step"]
    # nothing called `frac` may exceed 1 anywhere in the line
    def walk(o, path=""):
        if isinstance(o, dict):
            for k, v in o.items():
                if k == "frac" or k.endswith("_frac"):
                    assert v is None or v <= 1.0, (path, k, v)
                walk(v, path + "/" + k)
    walk(j)


def test_line_sheds_weight_rather_than_outgrow_the_driver():
    full = _canned()
    # forty workloads riding along: the emitter trims the secondaries, never the headline
    full["secondary"] = {f"{nm}_{i}": o for i in range(8) for nm, o in full["secondary"].items()}
    line = bench.compact_line(bench._finite(full))
    assert len(line) <= bench.LINE_LIMIT
    j = json.loads(line)
    assert "roofline" in j and "cpu_baseline" in j and j["value"] > 0


def test_nan_and_numpy_values_do_not_reach_the_line():
    import numpy as np
    full = _canned()
    full["planted_recall"] = float("nan")
    full["hits_per_step"] = np.float64(12.5)
    full["roofline"]["finalize_kernels_ms"] = float("inf")
    j = json.loads(bench.compact_line(bench._finite(full)))
    assert "planted_recall" not in j and j["hits_per_step"] == 12.5 and j["roofline"]["finalize_kernels_ms"] is None
