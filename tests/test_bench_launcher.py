"""bench.py's own launcher and multi-rank plumbing on a box without a GPU (BENCH_MOCK=1 swaps the
HIP engine for a stand-in and the rendezvous runs over gloo): `python bench.py --gpus N` started
plainly must spawn its N ranks, shard the frames, bracket the timed region with barriers, push one
batch through dist.scatter_frames / gather_results and print exactly one JSON line."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*extra, n=2):
    env = dict(os.environ, BENCH_MOCK="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    wl = () if "--workload" in extra else ("--workload", "tiny")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), *wl,
                        "--steps", "2", "--warmup", "1", *extra], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout                    # ONE JSON line, nothing else on stdout
    return json.loads(lines[0])


def test_plain_start_with_two_gpus_launches_its_own_ranks():
    d = _run()
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["warmup"] == 1 and d["scaling"] == "weak"
    assert d["config"]["world_size"] == 2 and d["config"]["frames_per_rank"] == [2, 2]
    assert d["config"]["global_pairs_per_step"] == 4 and d["config"]["ingest"] == "resident"
    # the default run also times the rank-0 ingest pipeline (scatter -> compute -> gather) after the timed region
    assert d["ingest_rank0"]["ok"] is True and d["ingest_rank0"]["frames_per_step"] == 4 and d["ingest_rank0"]["pairs_per_s"] > 0
    # ... on the timed region's pairs per rank, once with dense XYZ images and once with the compacted point lists
    assert d["ingest_rank0"]["pairs_per_rank"] == d["config"]["pairs_per_gpu_per_step"]
    assert d["ingest_rank0"]["dense"]["ok"] is True and d["ingest_rank0"]["compact"]["ok"] is True
    assert d["ingest_rank0"]["compact"]["bytes_per_pair_gathered"] < d["ingest_rank0"]["dense"]["bytes_per_pair_gathered"]
    assert "BENCH_MOCK" in d["data"]
    for k in ("metric", "value", "unit", "ms_per_step", "higher_is_better", "vs_baseline", "dtype", "roofline"):
        assert k in d


def test_rank0_ingest_mode_three_ranks():
    d = _run("--ingest", "rank0", n=3)
    assert d["n_gpus"] == 3 and d["config"]["ingest"] == "rank0"
    assert d["config"]["frames_per_rank"] == [2, 2, 2]
    assert "rank 0 scatters" in d["config"]["parallelism"]


def test_rank0_ingest_mode_with_compacted_points():
    d = _run("--ingest", "rank0", "--xyz", "compact", n=2)
    assert d["n_gpus"] == 2 and d["config"]["ingest"] == "rank0"


def test_a_stalled_ingest_leg_fails_the_run():
    """the watchdog of the N > 1 ingest leg prints the finished headline line and ends the run with a non-zero status"""
    env = dict(os.environ, BENCH_MOCK="1", BENCH_INGEST_TIMEOUT="0.0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--workload", "tiny", "--steps", "2", "--warmup", "1"],
                       capture_output=True, text=True, env=env, timeout=600)
    lines = [l for l in r.stdout.splitlines() if l.strip().startswith("{")]
    if r.returncode == 0:      # (the leg finished before the watchdog's first look: nothing stalled, nothing to assert)
        assert json.loads(lines[0])["ingest_rank0"]["ok"] is True
    else:
        assert len(lines) == 1 and "timed out (watchdog)" in json.loads(lines[0])["ingest_rank0"]["error"]


def test_throughput_mode_workload_through_the_batch_entry():
    env_n = 2
    d = _run("--workload", "tinyt", n=env_n)
    assert d["config"]["batch_entry"] is True and d["config"]["pairs_per_gpu_per_step"] == 3
    assert d["config"]["frames_per_rank"] == [3, 3] and d["ingest_rank0"]["ok"] is True
    assert "floor_bytes" in d and "unfused_model_bytes" in d and "frac_whole_step" not in d


def test_single_gpu_path_needs_no_launcher():
    d = _run(n=1)
    assert d["n_gpus"] == 1 and d["config"]["world_size"] == 1 and "ingest_rank0" not in d


def test_without_the_mock_and_without_a_gpu_it_fails_loudly():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    env = {k: v for k, v in os.environ.items() if k not in ("BENCH_MOCK", "RANK", "WORLD_SIZE")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode != 0 and "needs a GPU" in (r.stderr + r.stdout)
