"""What a TRIVIAL kernel gets for the fused upsampler's byte pattern on this GPU: 16.8 MB read + 33.5 MB written per launch (the 1024-px
geometry at batch 8), rotating over buffers beyond the Infinity Cache, launches captured in a HIP graph.  torch's own elementwise kernels
(a bf16 copy of 25 MB = 25 read + 25 written, and x -> cat(x, x) = 16.8 read + 33.5 written).  python scripts/hbm_copy_floor.py"""
import json, torch
dev = torch.device("cuda:0")
def timed(fn, per=24, reps=40):
    for i in range(5): fn(i)
    torch.cuda.synchronize()
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for i in range(per): fn(i)
    torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (reps * per)
nb = 12
a = [torch.randn(8 * 4096 * 256 * 3 // 2, device=dev).to(torch.bfloat16) for _ in range(nb)]       # 25.2 MB each
b = [torch.empty_like(x) for x in a]
us = timed(lambda i: b[i % nb].copy_(a[i % nb]))
print(json.dumps({"pattern": "copy 25.2 MB -> 25.2 MB (50.3 MB moved)", "us": round(us, 2), "TBps": round(50.33e6 / us / 1e6, 2)}))
x = [torch.randn(8 * 4096 * 256, device=dev).to(torch.bfloat16) for _ in range(nb)]               # 16.8 MB each
y = [torch.empty(2, 8 * 4096 * 256, dtype=torch.bfloat16, device=dev) for _ in range(nb)]           # 33.5 MB each
def dup(i):
    y[i % nb][0].copy_(x[i % nb]); y[i % nb][1].copy_(x[i % nb])
us2 = timed(dup)
print(json.dumps({"pattern": "16.8 MB read (twice, 2nd from cache) -> 33.5 MB written, two launches", "us": round(us2, 2), "TBps": round(50.33e6 / us2 / 1e6, 2)}))
z = [torch.empty(8 * 4096 * 256 * 2, dtype=torch.bfloat16, device=dev) for _ in range(nb)]
us3 = timed(lambda i: z[i % nb].fill_(1.0))
print(json.dumps({"pattern": "33.5 MB written only (fill)", "us": round(us3, 2), "TBps": round(33.55e6 / us3 / 1e6, 2)}))
