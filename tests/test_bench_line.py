"""The line bench.py prints is the driver's only view of a round: it must be ONE compact strict-JSON object below 4 KB carrying the contract's
keys, `roofline` and `cpu_baseline` -- whatever the full result object holds (the round-5 line grew to 30 KB and was not read)."""
import copy
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

CONTRACT_KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
                 "config", "roofline", "cpu_baseline")


def _strict_loads(line):
    def no_constants(x):
        raise ValueError("non-finite constant %r in the line" % x)
    return json.loads(line, parse_constant=no_constants)


@pytest.fixture(scope="module")
def canned():
    """a full result object as a default run produced it (round 5: 30 KB, every informational key)"""
    with open(os.path.join(ROOT, "profiles", "r05_bench_default.json")) as f:
        return json.loads(f.read().strip().splitlines()[-1])


def test_compact_line_is_small_strict_and_complete(canned):
    import bench
    line = bench.compact_line(canned, "bench_extras.json")
    assert "\n" not in line and len(line.encode()) < bench.LINE_MAX_BYTES <= 4096
    d = _strict_loads(line)
    for k in CONTRACT_KEYS:
        assert k in d, k
    assert d["value"] == canned["value"] and d["ms_per_step"] == canned["ms_per_step"] and d["steps"] == canned["steps"]
    assert d["config"]["workload"] == "kitti_shaped_1241x376_2000pts_ba10" and "model" not in d["config"]
    assert d["config"]["sequences_per_gpu"] == 256 and d["config"]["ba_lm_iteration_budget"] == 30
    assert d["config"]["stream_layout"] == {"layout": 2, "gate_groups": 4, "reserved_cus": 32} and d["config"]["solves_stopped_by_cap"] == 0
    r = d["roofline"]
    for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "avg_launch_us", "algorithmic_bytes_per_launch"):
        assert k in r, k
    assert r["bound"] == "hbm" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-4
    assert r["valu"]["frac"] > 0 and r["issue"]["frac_of_quarter_rate_capacity"] > 0
    assert len(r["kernels"]) == 3 and r["kernels"][0]["kernel"].startswith("k_klt_track")
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] == 16 and c["value"] > 0 and c["unit"] == "frames/s" and c["reference_recipe_ba_seconds_per_adjust"] > 0
    assert d["extras_file"] == "bench_extras.json"
    # the informational keys stay in the side file
    for k in ("pipeline_step", "single_sequence", "klt_only", "dropin_step", "lm_cap_10", "config5_n1"):
        assert k not in d


def test_compact_line_carries_the_second_figures_and_sheds_them_before_the_limit(canned):
    import bench
    full = copy.deepcopy(canned)
    full["host_frames"] = {"value": 44000.0, "unit": "frames/s", "ms_per_step": 5.8, "h2d_gb_s": 20.5}
    full["closed_loop_w10_256"] = {"value": 49871.0, "unit": "frames/s", "ms_per_step": 5.13, "sequences": 256, "capacity_policy_frames": 30720,
                                   "roofline": {"kernel": "k_klt_track", "frac": 0.70}}
    d = _strict_loads(bench.compact_line(full, "bench_extras.json"))
    assert d["host_frames"]["value"] == 44000.0 and d["closed_loop_w10_256"]["value"] == 49871.0
    # a second figure that explodes (say, an error text of 10 KB) is dropped, the contract's keys are not
    full["closed_loop_w10_256"] = {"error": "x" * 10000}
    line = bench.compact_line(full, "bench_extras.json")
    assert len(line.encode()) < bench.LINE_MAX_BYTES
    d = _strict_loads(line)
    assert "closed_loop_w10_256" not in d
    for k in CONTRACT_KEYS:
        assert k in d, k


def test_compact_line_never_prints_a_non_finite_number(canned):
    import bench
    import numpy as np
    full = copy.deepcopy(canned)
    full["roofline"]["traffic"] = float("nan")
    full["roofline"]["valu"]["frac"] = np.float32("inf")
    full["cpu_baseline"]["per_core"] = np.float64("nan")
    full["config"]["solves_note"] = np.int64(3)
    d = _strict_loads(bench.compact_line(full))
    assert d["roofline"]["traffic"] is None and "valu" not in d["roofline"] and "per_core" not in d["cpu_baseline"]


def test_compact_line_of_the_other_workloads():
    """config 5 and the closed loop print through the same function: no roofline.kernels, a null cpu_baseline"""
    import bench
    full = {"metric": "frames/sec, Pipeline.step resident on the device", "value": 49871.0, "unit": "frames/s", "n_gpus": 1, "steps": 40, "warmup": 10,
            "ms_per_step": 5.13, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8/i32 + f64", "data": "synthetic",
            "config": {"workload": "pipeline_step_closed_loop_1241x376_ba10", "sequences_per_gpu": 256, "batched_contexts_per_gpu": 1, "parallelism": "independent sequences"},
            "pipeline": {"big": list(range(5000))}, "roofline": {"bound": "hbm", "kernel": "k_klt_track", "achieved": 5611.3, "peak": 8000.0, "unit": "GB/s", "frac": 0.7014,
                                                                "traffic": None, "avg_launch_us": 2953.6, "algorithmic_bytes_per_launch": 16573466624},
            "cpu_baseline": None}
    line = bench.compact_line(full)
    d = _strict_loads(line)
    assert len(line) < 1500 and d["cpu_baseline"] is None and d["roofline"]["frac"] == 0.7014 and "pipeline" not in d
