"""bench.py's contract on a GPU: one JSON line on stdout with the agreed keys, and the
multi-process path (row strips, gather to rank 0) producing the same frame as one process."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu

REQUIRED = ["metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
            "vs_baseline", "dtype", "data", "config", "roofline"]


def run_bench(args, env_extra=None, launcher=None):
    env = dict(os.environ)
    env.update(env_extra or {})
    cmd = (launcher or [sys.executable]) + [os.path.join(ROOT, "bench.py")] + args
    p = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    lines = [l for l in p.stdout.decode().splitlines() if l.strip()]
    assert len(lines) == 1, lines          # exactly one line on stdout
    return json.loads(lines[0])


def test_single_process_line():
    d = run_bench(["--workload", "tiny", "--steps", "6", "--warmup", "2"])
    for k in REQUIRED + ["cpu_baseline"]:
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 6 and d["warmup"] == 2
    assert d["value"] > 0 and d["vs_baseline"] is None and d["dtype"] == "f32"
    rf = d["roofline"]
    for k in ["bound", "achieved", "peak", "unit", "frac", "traffic"]:
        assert k in rf, k
    assert rf["bound"] == "hbm" and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0


def test_two_ranks_render_the_same_frame():
    args = ["--workload", "tiny", "--steps", "6", "--warmup", "2", "--no-cpu-baseline"]
    one = run_bench(args)
    launcher = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                "--master-addr", "127.0.0.1", "--master-port", "29533"]
    two = run_bench(["--gpus", "2"] + args, env_extra={"RT_BENCH_BACKEND": "gloo", "RT_BENCH_SHARE_GPU": "1"},
                    launcher=launcher)
    assert two["n_gpus"] == 2
    assert two["config"]["frame_crc32"] == one["config"]["frame_crc32"]
