"""Writes the attention forward's output bits for a few shapes to a file (compare two builds / env settings with cmp)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from medplib_amd import ops
dev = torch.device("cuda:0")
torch.manual_seed(0)
outs = []
for (B, S, H, D, causal) in [(2, 639, 8, 128, True), (1, 1257, 8, 128, True), (2, 577, 8, 64, False), (1, 67, 2, 128, True)]:
    qkv = torch.randn(B, S, 3, H, D, device=dev).to(torch.bfloat16)
    outs.append(ops.attention(qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2], causal=causal).cpu())
    kvm = (torch.rand(B, S) > 0.2).to(torch.uint8); kvm[:, :4] = 1
    outs.append(ops.attention(qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2], causal=causal, key_valid=kvm.to(dev)).cpu())
torch.save(outs, sys.argv[1])
