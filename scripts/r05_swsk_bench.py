import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from medplib_amd import ops
dev = torch.device("cuda:0"); T, ff = 5112, 11008
g = torch.Generator(device=dev).manual_seed(0)
n = 3
mk = lambda *s: [torch.randn(*s, generator=g, device=dev).to(torch.bfloat16) for _ in range(n)]
dact, gu, dtd, tg = mk(T, ff), mk(T, 2 * ff), mk(T, 64), mk(T, 64)
ATd = mk(ff, 64)[0]; Bt = mk(64, 2 * ff)[0]
def timed(fn, reps=5):
    best = 1e9
    for _ in range(reps):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); s.record()
        for i in range(n): fn(i)
        e.record(); torch.cuda.synchronize(); best = min(best, s.elapsed_time(e) * 1e3 / n)
    return best
for p in (0.0, 0.05):
    def two(i):
        d = ops.lora_up_add_swiglu_bwd(dtd[i], ATd, dact[i], gu[i], 8, p, 7)
        ops.tn_skinny_down(d, tg[i], Bt, 16, 2.0, 2.0, reduce=False)
    a = timed(two)
    b = timed(lambda i: ops.swiglu_bwd_skinny(dtd[i], ATd, dact[i], gu[i], 8, p, 7, tg[i], Bt, 16, 2.0, 2.0, reduce=False))
    print(f"p={p}: two kernels {a:.1f} us, fused {b:.1f} us")
