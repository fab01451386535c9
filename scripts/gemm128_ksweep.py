"""Fixed cost of a 128x128-kernel launch at CLIP's o_proj / SAM-like shapes: time against K.  python scripts/gemm128_ksweep.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from medplib_amd import ops
dev = torch.device("cuda:0")
mk = lambda *s: torch.randn(*s, device=dev).to(torch.bfloat16)
ops.gemm_tile_policy(0)
for (M, N) in ((4616, 1024), (2048, 768), (2048, 3072)):
    for ep in ("", "bias+res"):
        pts = []
        for K in (64, 128, 256, 512, 1024, 2048):
            a = mk(M, K); ws = [mk(N, K) for _ in range(8)]
            out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
            kw = {"residual": mk(M, N), "bias": torch.randn(N, device=dev)} if ep else {}
            for i in range(3): ops.gemm(a, ws[i % 8], out=out, **kw)
            kern = ops.gemm_last_kernel()
            torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for i in range(40): ops.gemm(a, ws[i % 8], out=out, **kw)
            e.record(); torch.cuda.synchronize()
            us = s.elapsed_time(e) / 40 * 1e3
            pts.append((K, us))
            print(f"tile {kern} {M}x{N} {ep or 'plain':8s} K={K:5d}: {us:6.1f} us {2.0*M*N*K/us/1e6:7.1f} TF/s", flush=True)
ops.gemm_tile_policy(-1)
