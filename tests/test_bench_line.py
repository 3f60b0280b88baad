"""bench.py's ONE stdout line stays small and strictly parseable (the 23-KB line of round 5 came back
`parsed: null` from the driver).  The canned record is round 5's own full line (profiles/r05_bench_driver_flags.json)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402  (imports nothing that needs a GPU at module level)

CONTRACT = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
            "dtype", "data", "config", "roofline", "cpu_baseline")


def canned():
    text = open(os.path.join(ROOT, "profiles", "r05_bench_driver_flags.json")).read()
    rec = json.loads([l for l in text.splitlines() if l.lstrip().startswith("{")][-1])
    assert len(json.dumps(rec)) > 20000  # the record that did not parse
    return rec


def strict(text):
    def refuse(c):
        raise ValueError(c)
    return json.loads(text, parse_constant=refuse)  # NaN / Infinity are not JSON


def test_line_is_small_strict_json_with_every_contract_key():
    rec = canned()
    line = bench.compact_line(rec, "gpurun_out/bench_detail.json")
    assert "\n" not in line and len(line) < 6000, len(line)
    got = strict(line)
    for k in CONTRACT:
        assert k in got, k
    assert got["value"] == float(f"{rec['value']:.6g}") and got["unit"] == "edges/s" and got["dtype"] == "f32"
    rl = got["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in rl, k
    assert abs(rl["frac"] - rl["achieved"] / rl["peak"]) < 1e-4
    cb = got["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in cb, k
    assert set(got["config"]) <= {"workload", "graphs_per_step_per_gpu", "edges_per_step_per_gpu", "parallelism"}
    assert len(got["config"]["workload"]) <= 200
    # one number per side block
    assert got["sweep_whole"]["seconds"] > 0 and got["train"]["SGC-1"] > 0 and got["secondary"]["ms_per_step"] > 0
    assert all(len(v) == 2 for v in got["configs"].values())


def test_nan_and_oversized_blocks_cannot_break_the_line():
    rec = canned()
    rec["ms_per_step"] = float("nan")
    rec["configs"] = {f"config {i} " + "x" * 80: {"us": 1.0, "roofline": {"frac": 0.5}} for i in range(200)}
    line = bench.compact_line(rec)
    got = strict(line)
    assert len(line) < 6000 and got["ms_per_step"] is None and "configs" not in got
    for k in CONTRACT:
        assert k in got, k


def test_multi_gpu_record_keeps_the_measured_strong_speedup():
    rec = {k: v for k, v in canned().items() if k in CONTRACT and k != "cpu_baseline"}
    rec.update(n_gpus=8, sweep_whole={"jobs": 1680, "n_gpus": 8, "scaling": "strong", "seconds": 0.05, "one_gpu_s": 0.3,
                                      "strong_speedup_vs_1gpu": 6.0, "rows_equal_one_gpu": True, "per_rank_s": [0.05] * 8,
                                      "workload": "w" * 900})
    got = strict(bench.compact_line(rec))
    assert got["sweep_whole"]["strong_speedup_vs_1gpu"] == 6.0 and got["sweep_whole"]["rows_equal_one_gpu"] is True
    assert "workload" not in got["sweep_whole"] and "cpu_baseline" not in got
