"""The N > 1 code path of bench.py rehearsed on ONE GPU: two ranks share cuda:0 over gloo (`--backend gloo
--same-device`), cfg 2, strong and weak scaling.  Functional check only -- process group, frequency slabs, barrier /
all-reduce MAX timing, the all-gather of the maps and its record -- not a measurement (RCCL needs one GPU per rank)."""

import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("scaling,port", [("strong", 29541), ("weak", 29542)])
def test_bench_two_ranks_on_one_gpu(scaling, port):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--config", "2",
           "--backend", "gloo", "--same-device", "--scaling", scaling, "--pool-freqs", "16"]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    res = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-3000:]
    line = [l for l in res.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 2 and d["scaling"] == scaling and d["steps"] == 2 and d["value"] > 0
    nfreq_job = 64 if scaling == "strong" else 128
    assert d["allgather"]["frequencies_gathered"] == nfreq_job  # every rank ends with the whole map
    assert d["allgather"]["gathered_GB"] == pytest.approx(2 * d["allgather"]["shard_GB"])
    assert d["roofline"]["launches"] >= 2 and 0 < d["roofline"]["frac"] < 1.0
    assert d["cpu_baseline"] is None  # measured at N = 1 only
