import sys; sys.path.insert(0, "/root/repo")
import torch
from medplib_amd import ops
dev = torch.device("cuda:0")
T, d, E = 5112, 4096, 2
x = torch.randn(T, d, device=dev).to(torch.bfloat16); w = torch.rand(d, device=dev) + 0.5; wg = torch.randn(E, d, device=dev) * 0.02
def t(fn, n=50):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
print("rmsnorm", round(t(lambda: ops.rmsnorm(x, w, 1e-6)), 1), "us; rmsnorm_gate", round(t(lambda: ops.rmsnorm_gate(x, w, 1e-6, wg)), 1), "us")
