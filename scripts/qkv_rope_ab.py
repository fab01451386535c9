"""Fused qkv projection + RoPE at the decoder's shape: 256-row tiles (policy 0) against 320-row tiles (policy 2), bit-equality checked.
python scripts/qkv_rope_ab.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from medplib_amd import ops
dev = torch.device("cuda:0")
B, S, H, D, K = 8, 639, 32, 128, 4096
a = torch.randn(B * S, K, device=dev).to(torch.bfloat16)
ws = [(torch.randn(3 * H * D, K, device=dev) * 0.02).to(torch.bfloat16) for _ in range(6)]
inv = 1.0 / (10000.0 ** (torch.arange(0, D, 2, device=dev).float() / D))
ang = torch.arange(S + 4, device=dev).float()[:, None] * inv[None]
cos_t, sin_t = ang.cos().contiguous(), ang.sin().contiguous()
out = ops.padded_rows(B * S, 3 * H * D, dev)
res = {}
for pol in (0, 2, 0, 2):
    ops.gemm_tile_policy(pol)
    for i in range(3): ops.gemm_qkv_rope(a, ws[i % 6], cos_t, sin_t, S, H, D, out=out)
    kern = ops.gemm_last_kernel()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for i in range(30): ops.gemm_qkv_rope(a, ws[i % 6], cos_t, sin_t, S, H, D, out=out)
    e.record(); torch.cuda.synchronize()
    us = s.elapsed_time(e) / 30 * 1e3
    print(f"policy {pol} -> tile {kern}: {us:7.1f} us  {2.0 * B * S * 3 * H * D * K / us / 1e6:7.1f} TF/s", flush=True)
    res[pol] = ops.gemm_qkv_rope(a, ws[0], cos_t, sin_t, S, H, D).clone()
ops.gemm_tile_policy(1)
ops.gemm_qkv_rope(a, ws[0], cos_t, sin_t, S, H, D, out=out)
print("default policy picks tile", ops.gemm_last_kernel(), "| bit-equal:", torch.equal(res[0], res[2]))
ops.gemm_tile_policy(-1)
