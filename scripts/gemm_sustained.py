"""The decoder's four projection shapes run back to back for whole steps' worth of time (32 layers x 4 GEMMs, rotating weight buffers, as
in the model) with every launch bracketed by HIP events: what does each shape cost under the SUSTAINED clocks of a training step,
against the 30-launch bursts of scripts/gemm_sweep.py?  python scripts/gemm_sustained.py [steps] [weight sets] [residual 0|1]
(weight sets = 32: as many distinct weight buffers as the model has layers, 13 GB: page / TLB locality as in the model)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from medplib_amd import ops

dev = torch.device("cuda:0")
M = 5112
SH = {"qkv": (12288, 4096), "o": (4096, 4096), "gate|up": (22016, 4096), "down": (4096, 11008)}
NL = int(sys.argv[2]) if len(sys.argv) > 2 else 4  # distinct weight sets (> the 256 MB Infinity Cache together)
RES = len(sys.argv) > 3 and sys.argv[3] == "1"     # o / down with the residual epilogue, as in the decoder layer
mk = lambda *s: torch.randn(*s, device=dev).to(torch.bfloat16)
W = {k: [mk(n, kk) for _ in range(NL)] for k, (n, kk) in SH.items()}
A = {4096: mk(M, 4096), 11008: mk(M, 11008)}
R = mk(M, 4096)
O = {k: torch.empty(M, n, dtype=torch.bfloat16, device=dev) for k, (n, kk) in SH.items()}
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 6
ev = []
torch.cuda.synchronize()
for st in range(steps):
    for l in range(32):
        for k, (n, kk) in SH.items():
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            ops.gemm(A[kk], W[k][l % NL], out=O[k], residual=(R if RES and n == 4096 else None))
            e.record()
            ev.append((st, k, s, e))
torch.cuda.synchronize()
for st in range(steps):
    line = []
    tot = 0.0
    for k, (n, kk) in SH.items():
        t = [s.elapsed_time(e) * 1e3 for (s_, k_, s, e) in ev if s_ == st and k_ == k]
        us = sum(t) / len(t)
        tot += sum(t)
        line.append(f"{k} {us:7.1f} us {2.0 * M * n * kk / us / 1e6:7.1f} TF/s")
    print(f"step {st}: " + " | ".join(line) + f" | 128 launches {tot / 1e3:6.2f} ms", flush=True)
