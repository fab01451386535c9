"""The two MoE expert projections of a 7B-MoE layer (E = 2, 5112 tokens, capacity 3834) on 256x256 tiles (policy 0) and 320x256 tiles (policy 2):
gate|up with the dispatch gather + SwiGLU pairing, down with the combine scatter.  python scripts/expert_gemm_ab.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from medplib_amd import ops
dev = torch.device("cuda:0")
E, T, d, ff, cap = 2, 5112, 4096, 11008, 3834
x = torch.randn(T, d, device=dev).to(torch.bfloat16)
w_gu = [(torch.randn(E, 2 * ff, d, device=dev) * 0.02).to(torch.bfloat16) for _ in range(2)]
w_dn = [(torch.randn(E, d, ff, device=dev) * 0.02).to(torch.bfloat16) for _ in range(2)]
weight = torch.rand(T, device=dev)
for c0 in (2556, 2500, 2400):
    counts = torch.tensor([c0, T - c0], dtype=torch.int32, device=dev)
    perm = torch.randperm(T, device=dev).int()
    slot_token = torch.zeros(E, cap, dtype=torch.int32, device=dev)
    slot_token[0, :c0] = perm[:c0]; slot_token[1, :T - c0] = perm[c0:]
    act = torch.empty(E, cap, ff, dtype=torch.bfloat16, device=dev)
    out = torch.empty(T, d, dtype=torch.bfloat16, device=dev)
    for name, fn, flops in (("gate|up (gather, swiglu)", lambda i: ops.gemm_batched_rows(x, w_gu[i % 2], act, counts, a_rows=slot_token, act=ops.ACT_SWIGLU_PAIR, rows_stride=cap), 2.0 * T * 2 * ff * d),
                            ("down (combine)", lambda i: ops.gemm_batched_rows(act, w_dn[i % 2], out, counts, c_rows=slot_token, c_scale=weight, residual=x, rows_stride=cap), 2.0 * T * d * ff)):
        line = f"counts {c0}/{T - c0} {name:26s}"
        for pol in (0, 2):
            ops.gemm_tile_policy(pol)
            for i in range(3): fn(i)
            kern = ops.gemm_last_kernel()
            torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for i in range(20): fn(i)
            e.record(); torch.cuda.synchronize()
            us = s.elapsed_time(e) / 20 * 1e3
            line += f" | tile {kern}: {us:7.1f} us {flops / us / 1e6:7.1f} TF/s"
        print(line, flush=True)
ops.gemm_tile_policy(-1)
