"""The one line `bench.py` prints must be something the driver can read: it keeps about 8 KB of stdout, and round 4's
22 KB line (five dense-day records inside `extra`) left `BENCH_r04.json` with `parsed: null`.  Here the compact line is
built from RECORDED full records (the builder's own copies of round 4's lines under profiles/) -- no GPU."""

import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import bench  # noqa: E402

CONTRACT = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"}


def _load(name):
    with open(os.path.join(ROOT, "profiles", name)) as fh:
        return json.loads([ln for ln in fh.read().splitlines() if ln.startswith("{")][-1])  # (some records keep the run's other stdout lines)


@pytest.mark.parametrize("name", ["r04_bench_cfg3_rank_stop.json", "r04_bench_cfg3.json", "r03_bench_cfg3.json", "r04_bench_ml_cfg3_day_band_spread.json",
                                  "r04_bench_wiener_cfg3_day.json", "r04_bench_rehearsal_2ranks_selflaunch.json"])
def test_compact_line_fits_and_parses(name):
    out = _load(name)
    line = bench.compact_line(out)
    assert len(line) < 4096 and "\n" not in line
    d = json.loads(line)
    assert CONTRACT <= set(d)
    assert d["metric"] == out["metric"] and d["value"] == pytest.approx(out["value"], rel=1e-5) and d["ms_per_step"] == pytest.approx(out["ms_per_step"], rel=1e-5)
    assert d["n_gpus"] == out["n_gpus"] and d["steps"] == out["steps"] and d["warmup"] == out["warmup"] and d["scaling"] == out["scaling"]
    assert set(d["config"]) >= {"workload", "b_residency"} and all(len(v) <= 400 for v in d["config"].values() if isinstance(v, str))
    rf = d["roofline"]
    assert {"kernel", "bound", "achieved", "peak", "unit", "frac", "traffic"} <= set(rf)
    assert rf["frac"] == pytest.approx(rf["achieved"] / rf["peak"], rel=1e-4)
    assert d["extra_file"] == "bench_extra.json"


def test_compact_line_headline_record_carries_the_scalars():
    out = _load("r04_bench_cfg3_rank_stop.json")
    d = json.loads(bench.compact_line(out))
    # the dense makers' default day as four scalars each, not as 3 KB records
    for k in ("ml_day_s", "ml_gram_frac", "ml_stage1_hbm_frac", "wiener_day_s", "wiener_span_frac"):
        assert isinstance(d[k], float), k
    assert d["ml_day_s"] == pytest.approx(out["extra"]["ml_day"]["ms_per_step"] * 1e-3, rel=1e-5)
    assert d["wiener_span_frac"] == pytest.approx(out["extra"]["wiener_day"]["roofline"]["frac"], rel=1e-5)
    rf = d["roofline"]
    assert {"bytes_per_launch", "avg_launch_ms", "launches", "alone"} <= set(rf) and set(rf["alone"]) == {"frac", "avg_launch_ms"}
    assert rf["bytes_per_launch"] == out["roofline"]["bytes_per_launch"]
    cpu = d["cpu_baseline"]
    assert {"value", "unit", "cores", "kind", "blas", "cgroup_quota_cores", "loadavg", "sample"} <= set(cpu) and len(cpu["sample"]) <= 200
    assert "arms" not in cpu and "extra" not in d  # the per-arm table and the full secondary records stay in bench_extra.json
    assert d["stages_alone_ms"] == pytest.approx(out["stages_alone_ms"], rel=1e-5)
    assert d["secondary"]["many_days_D8"] == pytest.approx(out["extra"]["many_days"]["D8"]["value"], rel=1e-5)


def test_compact_line_survives_oversized_and_odd_records():
    out = _load("r04_bench_cfg3_rank_stop.json")
    out["config"]["workload"] = "w" * 5000
    out["cpu_baseline"]["sample"] = "s" * 5000
    out["dtype"] = "d" * 2000
    out["roofline"]["kernel"] = "k" * 2000
    out["extra"]["ml_day"] = {"error": "RuntimeError(" + "x" * 3000 + ")"}
    out["extra"]["error"] = "e" * 3000
    out["value"] = float("nan")  # json.dumps would print NaN, which is not JSON
    line = bench.compact_line(out)
    assert len(line) < 4096
    d = json.loads(line)
    assert d["value"] is None and "ml_day_error" in d and len(d["config"]["workload"]) == 400
    out["cpu_baseline"] = None  # (--no-cpu-baseline, ranks other than 0)
    assert json.loads(bench.compact_line(out))["cpu_baseline"] is None


def test_emit_prints_the_compact_line_last_and_writes_the_full_record(tmp_path, capsys, monkeypatch):
    out = _load("r04_bench_cfg3_rank_stop.json")
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    bench.emit(out)
    so = capsys.readouterr().out
    assert so.endswith("\n") and so.count("\n") == 1
    assert json.loads(so)["value"] == pytest.approx(out["value"], rel=1e-5)
    # what the driver does: the last 4 KB of stdout, last line
    assert json.loads(so[-4096:].splitlines()[-1])["metric"] == out["metric"]
    full = json.load(open(tmp_path / "bench_extra.json"))
    assert full["extra"]["ml_day"]["roofline"]["frac"] == pytest.approx(out["extra"]["ml_day"]["roofline"]["frac"], rel=1e-8)
    assert "arms" in full["cpu_baseline"]
    assert (tmp_path / "gpurun_out" / "bench_extra.json").exists()
