"""320-row tiles against 256-row tiles, shape by shape and epilogue by epilogue (mp_gemm_tile_policy 2 / 0), 40 launches each over rotating
weight buffers.  python scripts/gemm_tile_ab.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from medplib_amd import ops

dev = torch.device("cuda:0")
mk = lambda *s: torch.randn(*s, device=dev).to(torch.bfloat16)
CASES = [(5112, 4096, 4096, "res"), (5112, 4096, 11008, "res"), (5112, 4096, 22016, ""), (5112, 4096, 12288, ""), (5112, 12288, 4096, ""),
         (5112, 11008, 4096, ""), (4616, 4096, 1024, ""), (4616, 4096, 1024, "bias+quick_gelu"), (4616, 3072, 1024, "bias"), (4608, 4096, 4096, "bias")]
for M, N, K, ep in CASES:
    a = mk(M, K)
    ws = [mk(N, K) for _ in range(max(2, int(600e6 // (N * K * 2)) + 1))]
    out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    kw = {}
    if "res" in ep:
        kw["residual"] = mk(M, N)
    if "bias" in ep:
        kw["bias"] = torch.randn(N, device=dev)
    if "quick_gelu" in ep:
        kw["act"] = ops.ACT_QUICK_GELU
    line = f"{M:5d}x{N:5d}x{K:5d} {ep:16s}"
    for pol in (0, 2):
        ops.gemm_tile_policy(pol)
        for i in range(3):
            ops.gemm(a, ws[i % len(ws)], out=out, **kw)
        kern = ops.gemm_last_kernel()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for i in range(40):
            ops.gemm(a, ws[i % len(ws)], out=out, **kw)
        e.record(); torch.cuda.synchronize()
        us = s.elapsed_time(e) / 40 * 1e3
        line += f" | tile {kern}: {us:7.1f} us {2.0 * M * N * K / us / 1e6:7.1f} TF/s"
    print(line, flush=True)
ops.gemm_tile_policy(-1)
