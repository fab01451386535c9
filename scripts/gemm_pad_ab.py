import os, sys
sys.path.insert(0, "/root/repo")
import torch
from medplib_amd import ops
dev = torch.device("cuda:0")
mk = lambda *s: torch.randn(*s, device=dev).to(torch.bfloat16)
for M, N, K in [(5112, 4096, 12288), (5112, 4096, 4096), (5112, 4096, 8192), (5112, 4096, 11008)]:
    for pad in (0, 64, 192):
        a = mk(M, K + pad)[:, :K]
        ws = [mk(N, K + pad)[:, :K] for _ in range(4)]
        out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
        line = f"{M}x{N}x{K} pad {pad:3d}"
        for pol in (0, 2):
            ops.gemm_tile_policy(pol)
            for i in range(3): ops.gemm(a, ws[i % 4], out=out)
            torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for i in range(40): ops.gemm(a, ws[i % 4], out=out)
            e.record(); torch.cuda.synchronize()
            us = s.elapsed_time(e) / 40 * 1e3
            line += f" | {ops.gemm_last_kernel()}: {us:7.1f} us {2.0*M*N*K/us/1e6:7.1f} TF/s"
        print(line, flush=True)
