"""CPU: bench.py's multi-rank launch contract.  `python bench.py --gpus N` with no launcher in the environment must produce N ranks
by itself (re-exec under torch.distributed.run on 127.0.0.1) and print one JSON line whose n_gpus is N; a launcher that started
another number of ranks is refused.  `--dry` = gloo ranks and a stand-in gradient bucket through the engine's reduce path."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _clean_env():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["OMP_NUM_THREADS"] = "2"
    return env


def test_bench_gpus_2_self_launches_two_ranks():
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry", "--steps", "4", "--warmup", "1"],
                       env=_clean_env(), capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout                     # rank 0 only
    r = json.loads(lines[0])
    assert r["n_gpus"] == 2 and r["ranks"] == 2 and r["steps"] == 4 and r["warmup"] == 1 and r["bucket_sums_correct"] is True
    assert r["config"]["parallelism"] == "dp2" and r["dry"] is True


def test_bench_refuses_a_rank_count_other_than_requested():
    env = dict(_clean_env(), WORLD_SIZE="3", RANK="0", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry"], env=env, capture_output=True, text=True,
                       timeout=120)
    assert p.returncode != 0 and "refusing to report n_gpus" in (p.stderr + p.stdout)
    assert not [l for l in p.stdout.splitlines() if l.startswith("{")]
