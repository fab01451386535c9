"""A/B micro-benchmark of mp_gemm_bf16_nt on the shapes the 7B-MoE step issues (run on the GPU box).
MP_GEMM_VARIANT is read once per process, so each variant runs in its own process:  python scripts/gemm_bench.py [variant]"""
import os
import subprocess
import sys

SHAPES = [  # (name, M, N, K)
    ("llama qkv", 5112, 12288, 4096), ("llama o", 5112, 4096, 4096), ("expert gate|up (E=1 slice)", 2556, 22016, 4096),
    ("expert down", 2556, 4096, 11008), ("dense gate|up", 5112, 22016, 4096), ("dense down", 5112, 4096, 11008),
    ("clip fc1", 4616, 4096, 1024), ("clip qkv", 4616, 3072, 1024), ("sam qkv win", 6272, 2304, 768), ("square 4096", 4096, 4096, 4096),
    ("square 8192", 8192, 8192, 8192),
]


def run_variant(v):
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import torch
    from medplib_amd import ops
    dev = torch.device("cuda:0")
    print(f"variant {v}")
    for name, M, N, K in SHAPES:
        a = torch.randn(M, K, device=dev).to(torch.bfloat16); w = (torch.randn(N, K, device=dev) * 0.05).to(torch.bfloat16)
        out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
        for _ in range(3):
            ops.gemm(a, w, out=out)
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 20
        s.record()
        for _ in range(n):
            ops.gemm(a, w, out=out)
        e.record(); torch.cuda.synchronize()
        ms = s.elapsed_time(e) / n
        print(f"  {name:28s} {M:5d}x{N:5d}x{K:5d}  {ms * 1e3:8.1f} us  {2.0 * M * N * K / ms / 1e9:7.1f} TF/s", flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--child":
        run_variant(os.environ.get("MP_GEMM_VARIANT", "1"))
    else:
        for v in (sys.argv[1:] or ["1", "2"]):
            subprocess.run([sys.executable, __file__, "--child"], env=dict(os.environ, MP_GEMM_VARIANT=v))
