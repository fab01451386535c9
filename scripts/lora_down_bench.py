"""mp_lora_down_bf16 on the four shapes of a dense LoRA layer (stage-III targets gate/up/down, r = 8): bytes read / time.  python scripts/lora_down_bench.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from medplib_amd import ops
dev = torch.device("cuda:0")
T = 5112
for K, R, p, what in [(4096, 16, 0.05, "fwd gate|up: dropout(h2) A^T"), (11008, 8, 0.05, "fwd down: dropout(act) A^T"), (22016, 16, 0.0, "bwd gate|up: dY B"),
                      (4096, 8, 0.0, "bwd down: dY B"), (4096, 16, 0.0, "K 4096 no dropout")]:
    xs = [torch.randn(T, K, device=dev).to(torch.bfloat16) for _ in range(3)]
    A = torch.zeros(64, K, dtype=torch.bfloat16, device=dev); A[:R] = torch.randn(R, K, device=dev).to(torch.bfloat16)
    t = torch.empty(T, 64, dtype=torch.bfloat16, device=dev)
    xd = torch.empty(T, K, dtype=torch.bfloat16, device=dev) if p > 0 else None
    for i in range(3): ops.lora_down(xs[i % 3], A, t, R, p, 1, xd=xd)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for i in range(30): ops.lora_down(xs[i % 3], A, t, R, p, 1, xd=xd)
    e.record(); torch.cuda.synchronize()
    us = s.elapsed_time(e) / 30 * 1e3
    mb = T * K * 2 * (2 if p > 0 else 1) / 1e6
    print(f"{what:32s} K {K:5d} R {R:2d} p {p}: {us:7.1f} us, {mb:6.1f} MB -> {mb / us / 1e3 * 1e3:6.2f} TB/s", flush=True)
