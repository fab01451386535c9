"""Upper bound of the folded input norms, kernels alone (T = 5112, d = 4096, E = 2; inputs rotate over 24 buffers so nothing is cache-warm):
rmsnorm vs rstd-only, rmsnorm_gate vs its rstd form.  python scripts/r06_fold_bench.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from medplib_amd import ops
dev = torch.device("cuda:0")
T, d, E, NB = 5112, 4096, 2, 24
xs = [torch.randn(T, d, device=dev).to(torch.bfloat16) for _ in range(NB)]
outs = [torch.empty_like(xs[0]) for _ in range(NB)]
lnw = torch.rand(d, device=dev) + 0.5
wg = torch.randn(E, d, device=dev) * 0.02


def t(name, fn, n=240):
    for i in range(NB):
        fn(i)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for i in range(n):
        fn(i % NB)
    e.record(); torch.cuda.synchronize()
    print(f"{name:40s} {s.elapsed_time(e) / n * 1e3:7.2f} us", flush=True)


t("rmsnorm (writes h)", lambda i: ops.rmsnorm(xs[i], lnw, 1e-6, out=outs[i]))
t("rstd only", lambda i: ops.rmsnorm_gate_rstd(xs[i], lnw, 1e-6))
t("rmsnorm_gate (writes h)", lambda i: ops.rmsnorm_gate(xs[i], lnw, 1e-6, wg))
t("rmsnorm_gate_rstd", lambda i: ops.rmsnorm_gate_rstd(xs[i], lnw, 1e-6, wg))
r0, lg0, g0 = ops.rmsnorm_gate_rstd(xs[0], lnw, 1e-6, wg)
h1, lg1, g1 = ops.rmsnorm_gate(xs[0], lnw, 1e-6, wg)
print("gates bit-equal:", torch.equal(g0, g1), torch.equal(lg0, lg1), "rstd vs torch:", float((r0 - torch.rsqrt(xs[0].float().pow(2).mean(1) + 1e-6)).abs().max()))
