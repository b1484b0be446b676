"""Two ranks with the real engine on the one GPU of the box (needs an MI355X): `bench.py --gpus 2` started
plainly spawns its own ranks as CHILD processes (nothing here execs; the pytest process never hands its GPU
context to another program), BENCH_REHEARSE=1 lets both ranks share cuda:0 and rendezvous over gloo (RCCL
refuses two ranks on one device).  What is under test is the N > 1 code path with real kernels: sharding, the
timed region, and the rank-0 ingest pipeline (scatter / compute / gather, dist.IngestPipeline) whose gathered
results must equal the resident ones."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("workload", ["tiny", "tinyt"])
def test_two_ranks_share_the_gpu_and_the_ingest_pipeline_reproduces_the_resident_results(workload):
    env = dict(os.environ, BENCH_REHEARSE="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "BENCH_MOCK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--workload", workload, "--steps", "2",
                        "--warmup", "1", "--no-cpu-baseline"], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and "BENCH_REHEARSE" in d["data"]
    assert d["ingest_rank0"]["ok"] is True, d["ingest_rank0"]
    # both forms of the gather: dense XYZ images, and the compacted point lists (counts one step ahead of the points)
    assert d["ingest_rank0"]["dense"]["ok"] is True and d["ingest_rank0"]["compact"]["ok"] is True, d["ingest_rank0"]
    assert d["ingest_rank0"]["pairs_per_rank"] == d["config"]["pairs_per_gpu_per_step"]
    assert d["ingest_rank0"]["backend"] == "gloo" and d["ingest_rank0"]["frames_per_step"] == 2 * d["config"]["pairs_per_gpu_per_step"]
    assert d["config"]["batch_entry"] is (workload == "tinyt")
