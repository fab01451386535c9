"""Micro-benchmark of mp_attention_fwd_bf16 on the shapes of the step: Llama (B=8, H=32, S=639, D=128, causal), CLIP (S=577,
H=16, D=64), SAM window (25 windows x 8 images, S=196, H=12, D=64, rel-pos) and global (S=256).  python scripts/attn_bench.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from medplib_amd import ops

dev = torch.device("cuda:0")
CASES = [("llama causal", 8, 639, 32, 128, True, False), ("clip", 8, 577, 16, 64, False, False), ("sam window", 32, 196, 12, 64, False, True),
         ("sam global", 8, 256, 12, 64, False, True), ("llama S=1316 (config 5)", 4, 1316, 32, 128, True, False)]
for variant in (2, 0):
    print("variant", variant)
    for name, B, S, H, D, causal, relpos in CASES:
        qkv = torch.randn(B, S, 3, H, D, device=dev).to(torch.bfloat16)
        rel_h = rel_w = None
        if relpos:
            hw = int(S ** 0.5)
            rel_h = torch.randn(B * H, S, hw, device=dev); rel_w = torch.randn(B * H, S, hw, device=dev)
        out = torch.empty(B, S, H * D, dtype=torch.bfloat16, device=dev)
        for _ in range(3):
            ops.attention(qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2], out=out, causal=causal, rel_h=rel_h, rel_w=rel_w, variant=variant)
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 20
        s.record()
        for _ in range(n):
            ops.attention(qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2], out=out, causal=causal, rel_h=rel_h, rel_w=rel_w, variant=variant)
        e.record(); torch.cuda.synchronize()
        us = s.elapsed_time(e) * 1e3 / n
        fl = 4.0 * B * H * S * S * D * (0.5 if causal else 1.0)
        print(f"  {name:26s} {us:8.1f} us  {fl / us / 1e6:7.1f} TF/s", flush=True)

# backward (the decoder-backward piece for LoRA training): dQ + dK/dV kernels + delta
print("backward")
for name, B, S, H, D in [("llama causal", 8, 639, 32, 128), ("llama S=1316 (config 5)", 4, 1316, 32, 128)]:
    qkv = torch.randn(B, S, 3, H, D, device=dev).to(torch.bfloat16)
    d_out = torch.randn(B, S, H * D, device=dev).to(torch.bfloat16)
    out, lse2 = ops.attention_fwd_lse(qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2], causal=True)
    for _ in range(3):
        ops.attention_bwd(qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2], out, d_out, lse2, causal=True)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 20
    s.record()
    for _ in range(n):
        ops.attention_bwd(qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2], out, d_out, lse2, causal=True)
    e.record(); torch.cuda.synchronize()
    us = s.elapsed_time(e) * 1e3 / n
    fl = 2.0 * B * H * S * S * D * 0.5 * 7           # 3 matmuls in dQ, 4 in dK/dV, causal half
    print(f"  {name:26s} {us:8.1f} us  {fl / us / 1e6:7.1f} TF/s (incl. delta)", flush=True)
