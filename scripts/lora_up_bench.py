"""mp_lora_up_add_bf16 on the two shapes of a dense LoRA layer.  python scripts/lora_up_bench.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from medplib_amd import ops
dev = torch.device("cuda:0")
T = 5112
for K, R in [(4096, 16), (11008, 8), (4096, 8), (12288, 32)]:
    dt = torch.zeros(T, 64, dtype=torch.bfloat16, device=dev); dt[:, :R] = torch.randn(T, R, device=dev).to(torch.bfloat16)
    AT = torch.zeros(K, 64, dtype=torch.bfloat16, device=dev); AT[:, :R] = torch.randn(K, R, device=dev).to(torch.bfloat16)
    dxs = [torch.randn(T, K, device=dev).to(torch.bfloat16) for _ in range(3)]
    for i in range(3): ops.lora_up_add(dt, AT, dxs[i], R, 0.05, 3)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for i in range(30): ops.lora_up_add(dt, AT, dxs[i % 3], R, 0.05, 3)
    e.record(); torch.cuda.synchronize()
    us = s.elapsed_time(e) / 30 * 1e3
    print(f"K {K:5d} R {R:2d}: {us:6.1f} us, {T * K * 4 / 1e6 / us * 1e3 / 1e3:5.2f} TB/s (read + write)", flush=True)
