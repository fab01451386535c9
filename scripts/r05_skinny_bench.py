"""The adapter branch's HBM-bound kernels alone on the chip, at the LoRA stage-III shapes (T = 5112 rows, r = 8 fused gate|up = R 16 / down R 8):
us per launch and GB/s of the algorithmic bytes, cold operands (a ring of buffers larger than the Infinity Cache)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from medplib_amd import ops

dev = torch.device("cuda:0")
T = 5112
g = torch.Generator(device=dev).manual_seed(0)


def ring(shape, n):
    return [torch.randn(shape, generator=g, device=dev, dtype=torch.float32).to(torch.bfloat16) for _ in range(n)]


def timed(fn, n, reps=3):
    best = 1e9
    for _ in range(reps):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); s.record()
        for i in range(n):
            fn(i)
        e.record(); torch.cuda.synchronize()
        best = min(best, s.elapsed_time(e) * 1e3 / n)
    return best


res = {}
for name, N, R in (("dB_down: dy[T,4096]^T t", 4096, 8), ("dA_down: act[T,11008]^T dt", 11008, 8), ("dB_gu: d_gu[T,22016]^T t", 22016, 16), ("dA_gu: h2[T,4096]^T dt", 4096, 16)):
    n = max(4, int(600e6 // (T * N * 2)) + 1)
    xs = ring((T, N), n); ts = ring((T, 64), n)
    for p in (0.0, 0.05):
        if p > 0 and not name.startswith("dA"):
            continue
        us = timed(lambda i: ops.tn_skinny(xs[i % n], ts[i % n], R, 1.0, p, 1234, reduce=False), n)
        res[f"tn_skinny {name} p={p}"] = {"us": round(us, 1), "GBps": round(T * N * 2 / us / 1e3, 1)}
        if p > 0:
            kbs = [torch.randint(0, 256, (T, N // 8), dtype=torch.uint8, device=dev) for _ in range(n)]
            us = timed(lambda i: ops.tn_skinny(xs[i % n], ts[i % n], R, 1.0, p, 1234, reduce=False, keep_bits=kbs[i % n]), n)
            res[f"tn_skinny {name} p={p} mask bytes"] = {"us": round(us, 1), "GBps": round(T * N * 2 / us / 1e3, 1)}
            del kbs
    del xs, ts
for name, K, R in (("t_gu = h2[T,4096] A^T", 4096, 16), ("t_down = act[T,11008] A^T", 11008, 8), ("dt_down = dy[T,4096] B", 4096, 8), ("dt_gu = d_gu[T,22016] B", 22016, 16)):
    n = max(4, int(600e6 // (T * K * 2)) + 1)
    xs = ring((T, K), n); A = ring((16, K), 1)[0]; ts = [torch.empty((T, 64), dtype=torch.bfloat16, device=dev) for _ in range(n)]
    for p in (0.0, 0.05):
        if p > 0 and not name.startswith("t_"):
            continue
        us = timed(lambda i: ops.lora_down(xs[i % n], A, ts[i % n], R, p, 99), n)
        res[f"lora_down {name} p={p}"] = {"us": round(us, 1), "GBps": round(T * K * 2 / us / 1e3, 1)}
    del xs, ts
# the input-gradient pass of the down adapter with the SwiGLU backward behind it: d_act [T, 11008] + gu [T, 22016] -> d gate|up [T, 22016]
ff = 11008
n = 3
dacts = ring((T, ff), n); gus = ring((T, 2 * ff), n); dts = ring((T, 64), n); AT = ring((ff, 64), 1)[0]
for p in (0.0, 0.05):
    us = timed(lambda i: ops.lora_up_add_swiglu_bwd(dts[i % n], AT, dacts[i % n], gus[i % n], 8, p, 77), n)
    res[f"lora_up_add_swiglu_bwd p={p}"] = {"us": round(us, 1), "GBps": round(T * ff * 2 * 5 / us / 1e3, 1)}
kbs = [torch.randint(0, 256, (T, ff // 8), dtype=torch.uint8, device=dev) for _ in range(n)]
us = timed(lambda i: ops.lora_up_add_swiglu_bwd(dts[i % n], AT, dacts[i % n], gus[i % n], 8, 0.05, 77, keep_bits=kbs[i % n]), n)
res["lora_up_add_swiglu_bwd p=0.05 mask bytes"] = {"us": round(us, 1), "GBps": round(T * ff * 2 * 5 / us / 1e3, 1)}
del dacts, gus, dts, kbs
# beside them: torch's sum over the same bf16 bytes (a reduction kernel, not a tuned read)
for N in (4096, 11008, 22016):
    n = max(4, int(600e6 // (T * N * 2)) + 1)
    xs = ring((T, N), n)
    us = timed(lambda i: xs[i % n].sum(), n)
    res[f"read floor [T,{N}]"] = {"us": round(us, 1), "GBps": round(T * N * 2 / us / 1e3, 1)}
    del xs
for k, v in res.items():
    print(f"{k:50s} {v['us']:8.1f} us  {v['GBps']:8.1f} GB/s")
json.dump(res, open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "r05_skinny_bench.json"), "w"), indent=1)


# ---- the two fused passes of the backward against the kernels they replace (T = 5112)
def _timed_fixed(fn, n, reps=5):
    best = 1e9
    for _ in range(reps):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); s.record()
        for i in range(n):
            fn(i)
        e.record(); torch.cuda.synchronize(); best = min(best, s.elapsed_time(e) * 1e3 / n)
    return best


def fused_passes():
    d, R, n = 4096, 16, 6
    mk = lambda k, *s: [torch.randn(*s, generator=g, device=dev).to(torch.bfloat16) for _ in range(k)]
    xs, dys, adds, dts = mk(n, T, d), mk(n, T, d), mk(n, T, d), mk(n, T, 64)
    w = torch.ones(d, device=dev); AT = mk(1, d, 64)[0]
    kb = [torch.randint(0, 256, (T, d // 8), dtype=torch.uint8, device=dev) for _ in range(n)]
    for p in (0.0, 0.05):
        a_ = _timed_fixed(lambda i: ops.rmsnorm_bwd(xs[i], w, dys[i], 1e-5, add=adds[i]), n)
        b_ = _timed_fixed(lambda i: ops.lora_up_add(dts[i], AT, dys[i], R, p, 7), n)
        c_ = _timed_fixed(lambda i: ops.rmsnorm_bwd_up(xs[i], w, dys[i], 1e-5, dts[i], AT, R, p, 7, add=adds[i], keep_bits=kb[i] if p > 0 else None), n)
        print(f"rmsnorm_bwd + lora_up_add p={p}: {a_:.1f} + {b_:.1f} us; rmsnorm_bwd_up{' (mask bytes)' if p > 0 else ''} {c_:.1f} us")
    del xs, dys, adds, dts, kb
    ff, n = 11008, 3
    dact, gu, dtd, tg = mk(n, T, ff), mk(n, T, 2 * ff), mk(n, T, 64), mk(n, T, 64)
    ATd = mk(1, ff, 64)[0]; Bt = mk(1, 64, 2 * ff)[0]
    for p in (0.0, 0.05):
        def two(i):
            dg = ops.lora_up_add_swiglu_bwd(dtd[i], ATd, dact[i], gu[i], 8, p, 7)
            ops.tn_skinny_down(dg, tg[i], Bt, 16, 2.0, 2.0, reduce=False)
        a_ = _timed_fixed(two, n)
        b_ = _timed_fixed(lambda i: ops.swiglu_bwd_skinny(dtd[i], ATd, dact[i], gu[i], 8, p, 7, tg[i], Bt, 16, 2.0, 2.0, reduce=False), n)
        print(f"lora_up_add_swiglu_bwd + tn_skinny_down p={p}: {a_:.1f} us; swiglu_bwd_skinny {b_:.1f} us")


fused_passes()
