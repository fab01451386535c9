"""rmsnorm_bwd with the gate|up adapter's input-gradient term folded in, against the two kernels it replaces (T = 5112, d = 4096, R = 16)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from medplib_amd import ops
dev = torch.device("cuda:0"); T, d, R = 5112, 4096, 16
g = torch.Generator(device=dev).manual_seed(0)
n = 6
mk = lambda *s: [torch.randn(*s, generator=g, device=dev).to(torch.bfloat16) for _ in range(n)]
xs, dys, adds, dts = mk(T, d), mk(T, d), mk(T, d), mk(T, 64)
w = torch.ones(d, device=dev); AT = mk(d, 64)[0]
kb = [torch.randint(0, 256, (T, d // 8), dtype=torch.uint8, device=dev) for _ in range(n)]
def timed(fn, reps=5):
    best = 1e9
    for _ in range(reps):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); s.record()
        for i in range(n): fn(i)
        e.record(); torch.cuda.synchronize(); best = min(best, s.elapsed_time(e) * 1e3 / n)
    return best
for p in (0.0, 0.05):
    a = timed(lambda i: ops.rmsnorm_bwd(xs[i], w, dys[i], 1e-5, add=adds[i]))
    b = timed(lambda i: ops.lora_up_add(dts[i], AT, dys[i], R, p, 7))
    c = timed(lambda i: ops.rmsnorm_bwd_up(xs[i], w, dys[i], 1e-5, dts[i], AT, R, p, 7, add=adds[i]))
    c2 = timed(lambda i: ops.rmsnorm_bwd_up(xs[i], w, dys[i], 1e-5, dts[i], AT, R, p, 7, add=adds[i], keep_bits=kb[i])) if p > 0 else 0
    print(f"p={p}: rmsnorm_bwd {a:.1f} us, lora_up_add {b:.1f} us, fused {c:.1f} us, fused with mask bytes {c2:.1f} us")
