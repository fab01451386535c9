"""Attention forward against a float reference at sequence lengths beyond the unit tests' (plain causal, key padding left / right /
scattered, non-causal): max error per case and the rows that exceed 0.05.  python scripts/attn_check.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from medplib_amd import ops
dev = torch.device("cuda:0")
torch.manual_seed(0)
worst = 0.0
for (B, S, H, D) in [(1, 639, 4, 128), (1, 1257, 4, 128), (2, 1316, 4, 128), (1, 1257, 4, 64), (1, 2000, 2, 128), (3, 1087, 2, 128), (2, 775, 4, 128), (2, 769, 2, 128), (2, 129, 2, 64), (1, 65, 2, 128)]:
    qkv = torch.randn(B, S, 3, H, D, device=dev).to(torch.bfloat16)
    q, k, v = (qkv[:, :, i].float().permute(0, 2, 1, 3) for i in range(3))
    for causal in (True, False):
        for pad in ("none", "right", "left", "scattered"):
            kv = None
            if pad == "right":
                kv = torch.arange(S)[None, :] < torch.tensor([S - 97 * (i + 1) for i in range(B)])[:, None]
            elif pad == "left":
                kv = torch.arange(S)[None, :] >= torch.tensor([61 * (i + 1) for i in range(B)])[:, None]
            elif pad == "scattered":
                kv = torch.rand(B, S) > 0.3; kv[:, :3] = True
            out = ops.attention(qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2], causal=causal, key_valid=None if kv is None else kv.to(torch.uint8).to(dev))
            sc = q @ k.transpose(-1, -2) / D ** 0.5
            if causal:
                sc = sc.masked_fill(torch.triu(torch.ones(S, S, dtype=torch.bool, device=dev), 1), float("-inf"))
            if kv is not None:
                sc = sc.masked_fill(~kv.to(dev)[:, None, None, :], float("-inf"))
            p = torch.nan_to_num(sc.softmax(-1), nan=0.0)              # a row with no valid key: zero weights (HF eager semantics)
            ref = (p @ v).permute(0, 2, 1, 3).reshape(B, S, H * D)
            valid_rows = torch.ones(B, S, dtype=torch.bool, device=dev) if kv is None or not causal else (p.sum(-1).amax(1) > 0)
            err = ((out.float() - ref).abs().amax(dim=2) * valid_rows).amax()
            worst = max(worst, float(err))
            flag = "" if err < 0.05 else "   <-- BAD"
            print(f"B={B} S={S} D={D} causal={causal} pad={pad:9s} max err {float(err):.4f}{flag}", flush=True)
print("worst", worst)
assert worst < 0.05
