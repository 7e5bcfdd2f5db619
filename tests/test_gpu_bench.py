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
    # the gates the number is reported with: every primary ray's first hit and a few rows' radiance against the oracle
    pg = d["parity"]
    assert pg["primary_rays"] > 0 and pg["first_hit_id_mismatches"] == 0 and pg["first_hit_tuv_bit_mismatches"] == 0
    assert 0 < pg["primary_rays_that_hit"] <= pg["primary_rays"]
    # checked after the timed region, on the state after warm-up + timed iterations
    assert pg["radiance_iterations"] == 8 and pg["radiance_bit_equal"] and pg["radiance_rmse"] <= pg["radiance_rmse_tolerance"]
    assert d["parity_ok"] is True
    assert rf["traffic_source"] and "counter_frac" in rf


def test_repeats_report_the_median_region():
    d = run_bench(["--workload", "tiny", "--steps", "6", "--warmup", "2", "--repeats", "3", "--no-cpu-baseline"])
    c = d["config"]
    assert c["repeats"] == 3 and len(c["timed_regions_ms"]) == 3 and c["frame_after_iterations"] == 20
    assert abs(d["ms_per_step"] * 6 - sorted(c["timed_regions_ms"])[1]) < 1e-2
    assert d["roofline"]["launches"] == 3 and d["roofline"]["iterations_per_launch"] == 6
    assert d["parity"]["radiance_iterations"] == 20 and d["parity_ok"] is True


def test_two_ranks_render_the_same_frame():
    args = ["--workload", "tiny", "--steps", "6", "--warmup", "2", "--no-cpu-baseline"]
    one = run_bench(args)
    launcher = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                "--master-addr", "127.0.0.1", "--master-port", "29533"]
    two = run_bench(["--gpus", "2"] + args, env_extra={"RT_BENCH_BACKEND": "gloo", "RT_BENCH_SHARE_GPU": "1"},
                    launcher=launcher)
    assert two["n_gpus"] == 2
    assert two["config"]["frame_crc32"] == one["config"]["frame_crc32"]
    for d in (one, two):                   # rank 0 checks its own share: the whole frame, or its bands
        assert d["parity"]["first_hit_id_mismatches"] == 0 and d["parity"]["radiance_bit_equal"]
    assert two["parity"]["primary_rays"] * 2 == one["parity"]["primary_rays"]
    assert [k["rows"] for k in two["ranks"]] == [128, 128] and all(k["render_ms"] > 0 for k in two["ranks"])


def test_three_ranks_with_ragged_shares_render_the_same_frame():
    """256 rows are 32 bands: three ranks hold 11, 11 and 10 of them; the gathered frame is the single-rank one."""
    args = ["--workload", "tiny", "--steps", "6", "--warmup", "2", "--no-cpu-baseline"]
    one = run_bench(args)
    launcher = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "3",
                "--master-addr", "127.0.0.1", "--master-port", "29535"]
    three = run_bench(["--gpus", "3"] + args, env_extra={"RT_BENCH_BACKEND": "gloo", "RT_BENCH_SHARE_GPU": "1"},
                      launcher=launcher)
    assert three["n_gpus"] == 3 and [k["rows"] for k in three["ranks"]] == [88, 88, 80]
    assert three["config"]["frame_crc32"] == one["config"]["frame_crc32"]
    assert three["parity_ok"] is True


def test_plain_invocation_starts_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher around it starts two ranks itself (here both on the one GPU,
    over gloo) and prints the same frame checksum as one rank."""
    args = ["--workload", "tiny", "--steps", "6", "--warmup", "2", "--no-cpu-baseline"]
    one = run_bench(args)
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env.update({"RT_BENCH_BACKEND": "gloo", "RT_BENCH_SHARE_GPU": "1"})
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"] + args, cwd=ROOT, env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    lines = [l for l in p.stdout.decode().splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1, lines
    two = json.loads(lines[0])
    assert two["n_gpus"] == 2 and "ranks" in two["config"]["partition"]
    assert two["config"]["frame_crc32"] == one["config"]["frame_crc32"]


def test_more_gpus_than_visible_is_an_error():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "RT_BENCH_SHARE_GPU")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "64", "--workload", "tiny"], cwd=ROOT,
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
    assert p.returncode != 0 and not p.stdout.strip()
    assert b"GPU(s) are visible" in p.stderr


def test_short_timed_region_is_one_launch():
    """The command shape the round-end driver uses (--steps 20 --warmup 5): the timed region is ONE launch of 20
    iterations; the schedule calibration happened in the prepass."""
    d = run_bench(["--workload", "tiny", "--steps", "20", "--warmup", "5", "--no-cpu-baseline"])
    assert d["roofline"]["launches"] == 1 and d["roofline"]["iterations_per_launch"] == 20
    assert d["config"]["schedule_calibration_ms_per_iteration"] is not None
