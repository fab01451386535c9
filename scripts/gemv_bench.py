"""HBM roofline of mp_gemv_bf16 on the decode-step projections of the 7B model (cold weights: rotating buffers > Infinity Cache)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from medplib_amd import ops
dev = torch.device("cuda:0")
for name, N, K, act in (("qkv", 12288, 4096, 0), ("o", 4096, 4096, 0), ("gate|up", 22016, 4096, ops.ACT_SWIGLU_PAIR), ("down", 4096, 11008, 0), ("lm_head", 32267, 4096, 0)):
    nw = max(2, int(700e6 // (N * K * 2)) + 1)
    ws = [(torch.randn(N, K, device=dev) * 0.05).to(torch.bfloat16) for _ in range(nw)]
    x = torch.randn(1, K, device=dev).to(torch.bfloat16)
    for i in range(3):
        ops.gemv(x, ws[i % nw], act=act)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 40
    s.record()
    for i in range(n):
        ops.gemv(x, ws[i % nw], act=act)
    e.record(); torch.cuda.synchronize()
    us = s.elapsed_time(e) * 1e3 / n
    print(f"  gemv {name:8s} N={N:6d} K={K:6d}  {us:7.1f} us  {N * K * 2 / us / 1e3:7.1f} GB/s  ({N * K * 2 / us / 1e3 / 8000 * 100:.1f} % of 8 TB/s)", flush=True)
