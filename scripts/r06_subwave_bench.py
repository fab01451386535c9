"""The dense N = 4096 projections at 2556 rows (BASELINE configs[1], the dense VQA forward at batch 4: 128 tiles of 320 x 256 = half a wave) with and
without the sub-wave K split (MP_GEMM320_SUBWAVE, gemm320_bf16.hip: mp_gemm320_subwave_split) — each setting in its own process (the switch is read
once), cold weights, residual epilogue as in the decoder; then the whole configs[1] / configs[4] forwards (scripts/config_bench.py).
python scripts/r06_subwave_bench.py"""
import os
import subprocess
import sys

SHAPES = [("o_proj   B=4", 2556, 4096, 4096), ("down     B=4", 2556, 4096, 11008), ("o_proj   B=3", 1917, 4096, 4096), ("down     B=3", 1917, 4096, 11008),
          ("o_proj   B=8", 5112, 4096, 4096), ("down     B=8", 5112, 4096, 11008)]


def run():
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import torch
    from medplib_amd import ops
    dev = torch.device("cuda:0")
    for name, M, N, K in SHAPES:
        a = torch.randn(M, K, device=dev).to(torch.bfloat16)
        r = torch.randn(M, N, device=dev).to(torch.bfloat16)
        nw = max(2, int(600e6 // (N * K * 2)) + 1)
        ws = [(torch.randn(N, K, device=dev) * 0.02).to(torch.bfloat16) for _ in range(nw)]
        out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
        for i in range(3):
            ops.gemm(a, ws[i % nw], residual=r, out=out)
        torch.cuda.synchronize()
        ref = (a.float() @ ws[2 % nw].float().t()).to(torch.bfloat16).float() + r.float()
        err = (out.float() - ref).abs().max().item()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 30
        s.record()
        for i in range(n):
            ops.gemm(a, ws[i % nw], residual=r, out=out)
        e.record(); torch.cuda.synchronize()
        us = s.elapsed_time(e) / n * 1e3
        print(f"  {name}  {M} x {N} x {K}: {us:7.1f} us = {2 * M * N * K / us / 1e6:7.1f} TFLOP/s   kernel {ops.lib().raw('mp_gemm_last_kernel')()}   max err vs fp32 {err:.3f}")


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "run":
        run()
    else:
        for v in ("0", "1", "0", "1"):
            print(f"MP_GEMM320_SUBWAVE={v}", flush=True)
            subprocess.run([sys.executable, __file__, "run"], env=dict(os.environ, MP_GEMM320_SUBWAVE=v))
