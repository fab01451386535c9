"""Quick A/B of the 256x256 GEMM kernel variants (MP_GEMM_ABLATE is read once per process): correctness vs torch + timing on the
Llama shapes.  python scripts/gemm_ab.py  (on the GPU box)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from medplib_amd import ops

dev = torch.device("cuda:0")
SHAPES = [("qkv", 5112, 12288, 4096), ("o", 5112, 4096, 4096), ("gate|up", 5112, 22016, 4096), ("down", 5112, 4096, 11008),
          ("8192^3", 8192, 8192, 8192)]
PAD = int(os.environ.get("PAD", "0"))          # extra elements per row of A and W (leading-dimension padding experiment)
print("MP_GEMM_ABLATE =", os.environ.get("MP_GEMM_ABLATE", "0"), "SCHED =", os.environ.get("MP_GEMM_SCHED", "0"), "PAD =", PAD)
g = torch.Generator(device="cpu").manual_seed(0)
for M, N, K in [(5112, 4096, 4096), (700, 512, 192), (256, 256, 64)]:
    a = torch.randn(M, K, generator=g).to(torch.bfloat16).to(dev)
    w = (torch.randn(N, K, generator=g) * 0.05).to(torch.bfloat16).to(dev)
    out = ops.gemm(a, w)
    ref = a.float() @ w.float().t()
    err = (out.float() - ref).abs().max().item() / ref.abs().max().item()
    print(f"  check {M}x{N}x{K}: rel err {err:.2e}", flush=True)
for name, M, N, K in SHAPES:
    a = torch.randn(M, K + PAD, device=dev).to(torch.bfloat16)[:, :K]
    nw = max(2, int(600e6 // (N * K * 2)) + 1)
    ws = [(torch.randn(N, K + PAD, device=dev) * 0.05).to(torch.bfloat16)[:, :K] for _ in range(nw)]
    out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    for i in range(3):
        ops.gemm(a, ws[i % nw], out=out)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 24
    s.record()
    for i in range(n):
        ops.gemm(a, ws[i % nw], out=out)
    e.record(); torch.cuda.synchronize()
    ms = s.elapsed_time(e) / n
    print(f"  {name:8s} {ms * 1e3:8.1f} us  {2.0 * M * N * K / ms / 1e9:7.1f} TF/s", flush=True)
