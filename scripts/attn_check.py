import os, sys, torch
sys.path.insert(0, os.getcwd())
from medplib_amd import ops
dev = torch.device("cuda:0")
torch.manual_seed(0)
for (B, S, H, D) in [(1, 639, 4, 128), (1, 1257, 4, 128), (2, 1316, 4, 128), (1, 1257, 4, 64), (1, 2000, 2, 128)]:
    qkv = torch.randn(B, S, 3, H, D, device=dev).to(torch.bfloat16)
    out = ops.attention(qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2], causal=True)
    q, k, v = (qkv[:, :, i].float().permute(0, 2, 1, 3) for i in range(3))
    sc = q @ k.transpose(-1, -2) / D ** 0.5
    sc = sc.masked_fill(torch.triu(torch.ones(S, S, dtype=torch.bool, device=dev), 1), float("-inf"))
    ref = (sc.softmax(-1) @ v).permute(0, 2, 1, 3).reshape(B, S, H * D)
    err = (out.float() - ref).abs().amax(dim=(0, 2))
    bad = (err > 0.05).nonzero().flatten()
    print(S, D, "max err", float(err.max()), "bad rows", bad[:10].tolist(), "n bad", bad.numel(), "blocks", sorted(set((bad // 128).tolist()))[:12])
