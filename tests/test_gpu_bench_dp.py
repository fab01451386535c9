"""GPU: bench.py's data-parallel path with TWO real processes on one GPU.  RCCL refuses two ranks on one device, so the collectives
travel over gloo (MP_BENCH_BACKEND=gloo) and both ranks sit on cuda:0 (MP_BENCH_SHARE_GPU=1); everything else is the code the driver's
`torch.distributed.run --nproc-per-node N bench.py --gpus N` runs — rank environment, per-rank model and engine, the gradient bucket's
all-reduce on the communication stream beside the backward, barriers, the MAX over ranks, one JSON line from rank 0 — with real kernels
at 2 decoder layers.  (The RCCL side of the same path: a one-rank group in tests/test_gpu_bench_ep.py and MP_BENCH_FORCE_DIST=1.)"""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _launch(nproc, extra, env_extra=None):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update({"MP_BENCH_BACKEND": "gloo", "MP_BENCH_SHARE_GPU": "1", "OMP_NUM_THREADS": "4", "HSA_ENABLE_IPC_MODE_LEGACY": "0"})
    env.update(env_extra or {})
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", str(nproc), "--layers", "2", "--steps", "3", "--warmup", "1",
           "--roofline-steps", "0", "--no-cpu-baseline", "--no-lora-line", "--no-secondary"] + extra
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=1200)
    assert p.returncode == 0, p.stderr[-4000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]                     # rank 0 only
    return json.loads(lines[0])


def test_bench_two_ranks_share_one_gpu_data_parallel(dev):
    r = _launch(2, [])
    assert r["n_gpus"] == 2 and r["scaling"] == "weak" and r["config"]["parallelism"] == "dp2" and r["config"]["global_batch"] == 16
    assert r["steps"] == 3 and r["warmup"] == 1
    assert r["value"] > 0 and r["loss_last"] == r["loss_last"]
    # whole-job rate = global batch per MAX-over-ranks step time
    assert abs(r["value"] - 16 / (r["ms_per_step"] * 1e-3)) < 0.02 * r["value"]


def test_bench_two_ranks_share_one_gpu_expert_parallel(dev):
    """--ep 2: the two experts of BASELINE configs[4] on two ranks, the token exchange between two real processes (padded slabs)."""
    r = _launch(2, ["--ep", "2"])
    ep = r["ep"]
    assert r["n_gpus"] == 2 and ep["ep_size"] == 2 and ep["moe_layers"] == 2 and ep["exchanges_per_step"] == 4.0
    assert r["config"]["parallelism"] == "ep2 x dp1"
    assert r["value"] > 0 and r["loss_last"] == r["loss_last"]
