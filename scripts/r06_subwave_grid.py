"""Sub-wave dense launches (fewer tiles than CUs) per tiling: 256-row tiles, 320-row tiles whole, 320-row tiles with the half-wave split.
A launch that leaves CUs idle runs the busy ones faster (clock, L2 share), which the wave model of use_320 does not know: measure instead.
python scripts/r06_subwave_grid.py        (each tiling in its own process: the switches are read once)"""
import os
import subprocess
import sys

SHAPES = [(1278, 4096, 4096), (1278, 4096, 11008), (1917, 4096, 4096), (1917, 4096, 11008), (2556, 4096, 4096), (2556, 4096, 11008), (3195, 4096, 4096), (3195, 4096, 11008)]


def run():
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import torch
    from medplib_amd import ops
    dev = torch.device("cuda:0")
    pol = int(os.environ["POL"])
    for M, N, K in SHAPES:
        a = torch.randn(M, K, device=dev).to(torch.bfloat16)
        r = torch.randn(M, N, device=dev).to(torch.bfloat16)
        nw = max(2, int(600e6 // (N * K * 2)) + 1)
        ws = [(torch.randn(N, K, device=dev) * 0.02).to(torch.bfloat16) for _ in range(nw)]
        out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
        ops.gemm_tile_policy(pol)
        for i in range(3):
            ops.gemm(a, ws[i % nw], residual=r, out=out)
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 30
        s.record()
        for i in range(n):
            ops.gemm(a, ws[i % nw], residual=r, out=out)
        e.record(); torch.cuda.synchronize()
        us = s.elapsed_time(e) / n * 1e3
        print(f"  {M} x {N} x {K}: {us:7.1f} us = {2 * M * N * K / us / 1e6:7.1f} TFLOP/s   kernel {ops.gemm_last_kernel()}")


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "run":
        run()
    else:
        for name, env in (("256-row tiles", {"POL": "0"}), ("320-row tiles, whole", {"POL": "2", "MP_GEMM320_SUBWAVE": "0"}), ("320-row tiles, half-wave split", {"POL": "2"}),
                          ("256-row tiles", {"POL": "0"}), ("320-row tiles, whole", {"POL": "2", "MP_GEMM320_SUBWAVE": "0"})):
            print(name, flush=True)
            subprocess.run([sys.executable, __file__, "run"], env=dict(os.environ, **env))
