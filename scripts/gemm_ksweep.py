"""Fixed cost of a one-wave GEMM launch: time against K at 256 tiles (256-row kernel: 4096 x 4096, 320-row kernel: 5120 x 4096), plain and
with the residual epilogue -- the intercept of the line is launch + prologue + epilogue + drain.  python scripts/gemm_ksweep.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from medplib_amd import ops
dev = torch.device("cuda:0")
mk = lambda *s: torch.randn(*s, device=dev).to(torch.bfloat16)
for pol, M in ((0, 4096), (2, 5120)):
    N = 4096
    for ep in ("", "res"):
        pts = []
        for K in (128, 256, 512, 1024, 2048, 4096, 8192):
            a = mk(M, K); ws = [mk(N, K) for _ in range(max(2, int(600e6 // (N * K * 2)) + 1))][:24]
            out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
            kw = {"residual": mk(M, N)} if ep else {}
            ops.gemm_tile_policy(pol)
            for i in range(3): ops.gemm(a, ws[i % len(ws)], out=out, **kw)
            kern = ops.gemm_last_kernel()
            torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for i in range(40): ops.gemm(a, ws[i % len(ws)], out=out, **kw)
            e.record(); torch.cuda.synchronize()
            us = s.elapsed_time(e) / 40 * 1e3
            pts.append((K, us))
            print(f"tile {kern} M={M} {ep or 'plain':5s} K={K:5d}: {us:7.1f} us  {2.0*M*N*K/us/1e6:7.1f} TF/s", flush=True)
        (k1, t1), (k2, t2) = pts[-3], pts[-1]
        slope = (t2 - t1) / (k2 - k1)
        print(f"   -> slope {slope*64:.3f} us per 64-deep K step ({2.0*M*N*64/(slope*64)/1e6:.0f} TF/s in the loop), intercept {t1 - slope*k1:.1f} us", flush=True)
ops.gemm_tile_policy(-1)
