"""The N > 1 code path of bench.py rehearsed on ONE GPU: `python bench.py --gpus 2` starts its two ranks ITSELF (no
torchrun on the command line: the driver's command shape); they share cuda:0 over gloo (`--backend gloo
--same-device`), cfg 2, strong (the default) and weak scaling.  Functional check only -- process group, frequency slabs, barrier /
all-reduce MAX timing, the all-gather of the maps and its record -- not a measurement (RCCL needs one GPU per rank)."""

import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("scaling", [None, "weak"])
def test_bench_two_ranks_on_one_gpu(scaling):
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--config", "2",
           "--backend", "gloo", "--same-device", "--pool-freqs", "16", "--no-cpu-baseline"] + (["--scaling", scaling] if scaling else [])
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    res = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-3000:]
    line = [l for l in res.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    scaling = scaling or "strong"  # the N > 1 default: the metric's own job split over the ranks
    assert d["n_gpus"] == 2 and d["scaling"] == scaling and d["steps"] == 2 and d["value"] > 0
    assert d["ranks"]["ranks_seen_by_all_gather_into_tensor"] == 2 and d["launcher"]["ranks_started"] == 2
    nfreq_job = 64 if scaling == "strong" else 128
    assert d["allgather"]["frequencies_gathered"] == nfreq_job  # every rank ends with the whole map
    assert d["allgather"]["gathered_GB"] == pytest.approx(2 * d["allgather"]["shard_GB"], rel=1e-4)  # (the line carries 6 significant digits)
    assert d["roofline"]["launches"] >= 2 and 0 < d["roofline"]["frac"] < 1.0
    assert d["cpu_baseline"] is None  # (--no-cpu-baseline; otherwise the launcher times it before it starts the ranks)


def test_bench_refuses_more_ranks_than_gpus():
    """`--gpus 8` on a one-GPU box must fail loudly, not run one rank and print n_gpus = 1."""
    import torch

    if torch.cuda.device_count() >= 8:  # (counting devices does not initialise the GPU)
        pytest.skip("an 8-GPU box: the command is a real run here")
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--no-cpu-baseline"], cwd=ROOT,
                         env={k: v for k, v in os.environ.items() if k != "WORLD_SIZE"}, capture_output=True, text=True, timeout=300)
    assert res.returncode != 0 and "GPU(s) visible" in res.stderr and not res.stdout.strip()


def test_bench_default_command_shape_prints_one_short_parsable_last_line():
    """What round 4 got wrong: the driver keeps the tail of stdout and parses the LAST line.  The default command shape (N = 1,
    extras and CPU baseline on) at cfg 2 with a short CPU budget: the last line of stdout is JSON, under 4 KB, carries the
    contract's keys, `roofline` and `cpu_baseline`, and names the file the full record went to -- which exists and holds more."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--config", "2", "--cpu-seconds", "4", "--pool-freqs", "16"]
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    res = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stderr[-3000:]
    tail = res.stdout[-4096:]
    line = tail.splitlines()[-1]
    assert len(line) < 4096 and res.stdout.rstrip("\n").endswith(line)
    d = json.loads(line)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["value"] > 0 and 0 < d["roofline"]["frac"] < 1
    assert d["cpu_baseline"]["value"] > 0 and d["cpu_baseline"]["repeats"] >= 1 and len(d["cpu_baseline"]["sample"]) <= 200
    full = json.load(open(os.path.join(ROOT, d["extra_file"])))
    assert full["value"] == pytest.approx(d["value"], rel=1e-5) and "arms" in full["cpu_baseline"] and "extra" in full
