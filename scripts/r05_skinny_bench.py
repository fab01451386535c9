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
