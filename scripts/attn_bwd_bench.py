"""Attention backward (dQ + dK/dV kernels) at the trunk's shape, and the forward beside it.  python scripts/attn_bwd_bench.py
MP_ATTN_BWD_STAGES=2 restores the two-stage ring (A/B)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from medplib_amd import ops
dev = torch.device("cuda:0")
B, S, H, D = 8, int(os.environ.get("S", 639)), 32, 128
buf = ops.padded_rows(B * S, 3 * H * D, dev)
buf.copy_(torch.randn(B * S, 3 * H * D, device=dev))
q5 = buf.unflatten(0, (B, S)).unflatten(1 + 1, (3, H, D))
q, k, v = q5[:, :, 0], q5[:, :, 1], q5[:, :, 2]
out, lse = ops.attention_fwd_lse(q, k, v, causal=True)
d_out = torch.randn_like(out)
def timeit(fn, n=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
print(f"S={S} fwd_lse: {timeit(lambda: ops.attention_fwd_lse(q, k, v, causal=True)):7.1f} us", flush=True)
print(f"S={S} bwd (delta + dQ + dK/dV): {timeit(lambda: ops.attention_bwd(q, k, v, out, d_out, lse, causal=True)):7.1f} us", flush=True)
r = ops.attention_bwd(q, k, v, out, d_out, lse, causal=True)
print("checksum", [float(t.float().abs().sum()) for t in r[:3]])
