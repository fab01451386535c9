"""Host cost of one C-ABI call by op (no synchronisation inside the loop; the queue is drained before and after): what a launch costs the issuing
thread in Python + ctypes + the library's launcher + hipLaunchKernel.  python scripts/r06_launch_cost.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from medplib_amd import ops
dev = torch.device("cuda:0")
x = torch.randn(64, 4096, device=dev).to(torch.bfloat16)
w = torch.randn(4096, 4096, device=dev).to(torch.bfloat16)
big = torch.randn(5112, 4096, device=dev).to(torch.bfloat16)
lnw = torch.ones(4096, device=dev)
out_s = torch.empty_like(x); out_b = torch.empty_like(big); out_g = torch.empty((5112, 4096), dtype=torch.bfloat16, device=dev)
lib = ops.lib()
raw_add3 = lib.raw("mp_add3_bf16")
st = torch.cuda.current_stream().cuda_stream


def t(name, fn, n=300):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"{name:58s} issue {1e6 * (t1 - t0) / n:7.1f} us/call   (with drain {1e6 * (t2 - t0) / n:7.1f})", flush=True)


t("raw ctypes mp_add3_bf16, 64 x 4096 (prebuilt args)", lambda: raw_add3(x.data_ptr(), x.data_ptr(), None, out_s.data_ptr(), x.numel(), st))
t("ops.add3 small (allocates its output)", lambda: ops.add3(x, x))
t("ops.add3 small, out=", lambda: ops.add3(x, x, out=out_s))
t("ops.rmsnorm 5112 x 4096, out=", lambda: ops.rmsnorm(big, lnw, 1e-6, out=out_b))
t("ops.gemm 64 x 4096 x 4096 (128-tile kernel)", lambda: ops.gemm(x, w), n=200)
t("ops.gemm 5112 x 4096 x 4096 (320-row kernel), out=", lambda: ops.gemm(big, w, out=out_g), n=60)
t("torch.empty((5112, 4096), bf16)", lambda: torch.empty((5112, 4096), dtype=torch.bfloat16, device=dev))
t("torch.cuda.current_stream().cuda_stream", lambda: torch.cuda.current_stream().cuda_stream, n=2000)
