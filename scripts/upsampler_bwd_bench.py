"""The fused upsampler's training form (forward with the hypernetwork product + the recomputing backward and its two weight-gradient GEMMs)
at the model's geometry (16 x 16 tokens, batch 8) and at the 1024-px geometry (64 x 64): run under rocprofv3 --kernel-trace --stats
for the per-kernel times.  python scripts/upsampler_bwd_bench.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from medplib_amd.model import autograd_ops as A
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(1)
w1 = (torch.randn(256, 64, 2, 2, generator=g) * 0.06).to(dev).requires_grad_(); b1 = torch.zeros(64, device=dev, requires_grad=True)
lnw = torch.ones(64, device=dev, requires_grad=True); lnb = torch.zeros(64, device=dev, requires_grad=True)
w2 = (torch.randn(64, 32, 2, 2, generator=g) * 0.12).to(dev).requires_grad_(); b2 = torch.zeros(32, device=dev, requires_grad=True)
for G in (16, 64):
    n = 8
    src = torch.randn(n, G * G, 256, generator=g).to(dev).requires_grad_()
    hyper = (torch.randn(n, 32, generator=g) * 0.5).to(dev).requires_grad_()
    dm = torch.randn(n, 4 * G, 4 * G, generator=g).to(dev)
    for it in range(12):
        out = A.FusedUpsampleMaskFn.apply(src, w1, b1, lnw, lnb, w2, b2, hyper, G, 1e-6)
        out.backward(dm)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for it in range(20):
        out = A.FusedUpsampleMaskFn.apply(src, w1, b1, lnw, lnb, w2, b2, hyper, G, 1e-6)
        out.backward(dm)
    e.record(); torch.cuda.synchronize()
    print(f"grid {G} x {G}, batch {n}: forward + backward {s.elapsed_time(e) / 20 * 1e3:.1f} us per call (launch-bound wall time)", flush=True)
