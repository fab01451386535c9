"""Does the row stride of the fused qkv buffer matter to the attention kernels?  3 * 4096 * 2 B = 24 KiB rows put every token's q / k / v at the
same offset modulo 8 KiB (K = 12288 GEMM operands lose 12-14 % to that, scripts/gemm_pad_ab.py).  python scripts/attn_pad_ab.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from medplib_amd import ops
dev = torch.device("cuda:0")
B, S, H, D = 8, 639, 32, 128
for pad in (0, 64, 128, 320):
    buf = torch.randn(B * S, 3 * H * D + pad, device=dev).to(torch.bfloat16)
    q5 = buf[:, :3 * H * D].unflatten(0, (B, S)).unflatten(2, (3, H, D))
    q, k, v = q5[:, :, 0], q5[:, :, 1], q5[:, :, 2]
    for name, fn in (("fwd", lambda: ops.attention(q, k, v, causal=True, variant=0)), ("fwd_lse", lambda: ops.attention_fwd_lse(q, k, v, causal=True))):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(50): fn()
        e.record(); torch.cuda.synchronize()
        print(f"pad {pad:3d} {name}: {s.elapsed_time(e) / 50 * 1e3:7.1f} us", flush=True)
