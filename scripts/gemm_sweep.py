"""Shape sweep of the 256x256 bf16 GEMM: why do the Llama shapes (M = 5112, K = 4096) run below the 8192^3 rate?  Separates wave
quantisation (tiles / 256 CUs), K depth (prologue / epilogue share) and data-dependent clocks.  python scripts/gemm_sweep.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from medplib_amd import ops

dev = torch.device("cuda:0")
SHAPES = [(8192, 8192, 8192), (8192, 8192, 4096), (8192, 8192, 2048), (4096, 16384, 4096), (4096, 12288, 4096), (5120, 12288, 4096), (5112, 12288, 4096),
          (5112, 12288, 8192), (4096, 4096, 4096), (5112, 4096, 4096), (5112, 4096, 11008), (4096, 4096, 11008), (5112, 22016, 4096), (4096, 22016, 4096),
          (2556, 22016, 4096), (2560, 22016, 4096)]
for fill in ("randn", "zeros"):
    print("fill", fill)
    for M, N, K in SHAPES:
        mk = (lambda *s: torch.randn(*s, device=dev).to(torch.bfloat16)) if fill == "randn" else (lambda *s: torch.zeros(*s, dtype=torch.bfloat16, device=dev))
        a = mk(M, K)
        nw = max(2, int(600e6 // (N * K * 2)) + 1)
        ws = [mk(N, K) for _ in range(nw)]
        out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
        for i in range(3):
            ops.gemm(a, ws[i % nw], out=out)
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 30
        s.record()
        for i in range(n):
            ops.gemm(a, ws[i % nw], out=out)
        e.record(); torch.cuda.synchronize()
        ms = s.elapsed_time(e) / n
        tiles = -(-M // 256) * -(-N // 256)
        print(f"  {M:5d}x{N:5d}x{K:5d} tiles {tiles:5d} = {tiles / 256:5.2f} waves  {ms * 1e3:8.1f} us  {2.0 * M * N * K / ms / 1e9:7.1f} TF/s", flush=True)
        del ws, a, out
    if len(sys.argv) > 1 and sys.argv[1] == "--randn-only":
        break
