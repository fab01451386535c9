"""mp_moe_route_top1 at the decoder's token count.  python scripts/route_bench.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from medplib_amd import ops
dev = torch.device("cuda:0")
T, E = 5112, 2
gates = torch.softmax(torch.randn(T, E, device=dev), dim=-1)
draws = torch.rand(T, E, device=dev)
for cap, what in ((3834, "capacity 1.5 x (no expert over capacity)"), (2000, "capacity 2000 (draw-based selection runs)")):
    for _ in range(3): ops.moe_route_top1(gates, cap, draws, want_slot_token=True)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(50): ops.moe_route_top1(gates, cap, draws, want_slot_token=True)
    e.record(); torch.cuda.synchronize()
    print(f"{what}: {s.elapsed_time(e) / 50 * 1e3:.1f} us (incl. the wrapper's 6 small allocations)")
