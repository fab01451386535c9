"""HBM-roofline micro-benchmark of the fused mask-decoder upsampler (K14) at batch 8 for both geometries:
256-px SAM (16x16 tokens, 3.29 MB algorithmic) and 1024-px SAM (64x64 tokens, 50.5 MB).  Launches are captured in a HIP graph
so the time per iteration is not host-launch bound.  Prints one JSON line per geometry."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from medplib_amd import ops

HBM_PEAK = 8.0e12
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
w1 = torch.randn(256, 64, 2, 2, device=dev, generator=g) * 0.06
w2 = torch.randn(64, 32, 2, 2, device=dev, generator=g) * 0.12
w1p, w2p = ops.pack_upsampler_weights(w1, w2)
b1 = torch.randn(64, device=dev, generator=g) * 0.05; lw = torch.ones(64, device=dev); lb = torch.zeros(64, device=dev)
b2 = torch.randn(32, device=dev, generator=g) * 0.05
for grid in (16, 64):
    B = 8
    # rotate over enough distinct inputs that the 256 MiB Infinity Cache cannot hold the working set of the large geometry
    n_buf = 1 if grid == 16 else 12
    srcs = [torch.randn(B, grid * grid, 256, device=dev, generator=g).to(torch.bfloat16) for _ in range(n_buf)]
    ups = [torch.empty(B, 32, 4 * grid, 4 * grid, dtype=torch.bfloat16, device=dev) for _ in range(n_buf)]

    def run(i):
        ops.lib().call("mp_mask_upsample_fused_bf16", srcs[i % n_buf].data_ptr(), w1p.data_ptr(), b1.data_ptr(), lw.data_ptr(), lb.data_ptr(),
                       w2p.data_ptr(), b2.data_ptr(), None, ups[i % n_buf].data_ptr(), None, B, grid, grid, 1e-6,
                       torch.cuda.current_stream().cuda_stream)
    for i in range(5):
        run(i)
    torch.cuda.synchronize()
    iters = 24
    graph = None
    try:
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, stream=s):
                for i in range(iters):
                    run(i)
        torch.cuda.synchronize()
    except Exception as e:          # graph capture unavailable: plain back-to-back launches
        graph = None
        print("graph capture failed:", repr(e), file=sys.stderr)
    # pre-heat: a few ms of dense GEMM so the measurement is not taken on idle (un-ramped) clocks
    ha = torch.randn(4096, 4096, device=dev).to(torch.bfloat16); hw = torch.randn(4096, 4096, device=dev).to(torch.bfloat16)
    for _ in range(30):
        ops.gemm(ha, hw)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 200
    e0.record()
    for _ in range(reps):
        if graph is not None:
            graph.replay()
        else:
            for i in range(iters):
                run(i)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / (reps * iters)
    tokens = B * grid * grid
    bytes_alg = tokens * 256 * 2 + (256 * 256 + 128 * 64) * 2 + tokens * 16 * 32 * 2
    flops = tokens * (256 * 256 + 4 * 64 * 128) * 2
    print(json.dumps({"kernel": "upsample_fused_kernel", "geometry": f"{grid}x{grid} tokens ({grid * 16}-px SAM), batch 8, bf16",
                      "us_per_launch": round(us, 2), "algorithmic_MB": round(bytes_alg / 1e6, 2), "achieved_GBps": round(bytes_alg / us / 1e3, 1),
                      "frac_of_8TBps": round(bytes_alg / (us * 1e-6) / HBM_PEAK, 4), "TFLOPs": round(flops / us / 1e6, 1),
                      "graph": graph is not None}), flush=True)
