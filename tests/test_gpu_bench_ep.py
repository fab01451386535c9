"""GPU: bench.py's expert-parallel line (`--ep`, BASELINE configs[4]) through the real model on one GPU — the ICL batch, the token compressor and
mask encoder, `enable_expert_parallel`, the exchange on a one-rank group through both transports, the `ep` object — at 2 decoder layers so it takes
seconds.  What more than one rank adds (group shapes, capacity agreement, the exchange itself) is covered on gloo by tests/test_bench_launch.py and
tests/test_host_logic.py; this keeps the GPU side of `bench.py --gpus N --ep E` from breaking unseen."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra, env_extra=None):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra or {})
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--ep", "1", "--layers", "2", "--steps", "2", "--warmup", "1",
                        "--roofline-steps", "0"] + extra, env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    return json.loads(lines[0])


def test_bench_ep_line_on_one_gpu(dev):
    r = _run([])
    assert r["n_gpus"] == 1 and r["config"]["parallelism"] == "ep1 x dp1" and r["config"]["seq_len"] == 1273 and r["config"]["global_batch"] == 4
    assert "configs[4]" in r["config"]["workload"]
    ep = r["ep"]
    assert ep["ep_size"] == 1 and ep["moe_layers"] == 2 and ep["exchanges_per_step"] == 4.0 and ep["variable_split"] is False
    assert r["loss_last"] == r["loss_last"] and r["value"] > 0                      # finite loss, a rate


def test_bench_ep_line_capi_variable_split_on_a_one_rank_rccl_group(dev):
    r = _run(["--ep-comm", "capi", "--ep-variable"], {"MP_BENCH_FORCE_DIST": "1", "MASTER_PORT": "29677"})
    ep = r["ep"]
    assert ep["variable_split"] is True and "mp_alltoall_tokens" in ep["transport"] and ep["exchanges_per_step"] == 4.0
    assert r["rccl_ranks"] == {"ncclCommCount": 1, "sum_of_ones": 1.0}
