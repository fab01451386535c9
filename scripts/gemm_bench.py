"""A/B micro-benchmark of mp_gemm_bf16_nt on the shapes the 7B-MoE step issues (run on the GPU box).
MP_GEMM_VARIANT is read once per process, so each variant runs in its own process:  python scripts/gemm_bench.py [variant]"""
import os
import subprocess
import sys

SHAPES = [  # (name, M, N, K)
    ("llama qkv", 5112, 12288, 4096), ("llama o", 5112, 4096, 4096), ("expert gate|up (E=1 slice)", 2556, 22016, 4096),
    ("expert down", 2556, 4096, 11008), ("dense gate|up", 5112, 22016, 4096), ("dense down", 5112, 4096, 11008),
    ("clip fc1", 4616, 4096, 1024), ("clip qkv", 4616, 3072, 1024), ("clip o", 4616, 1024, 1024), ("clip fc2", 4616, 1024, 4096),
    ("projector 0", 4608, 4096, 1024), ("projector 2", 4608, 4096, 4096),
    ("sam qkv win", 6272, 2304, 768), ("sam proj", 2048, 768, 768), ("sam fc1", 2048, 3072, 768), ("sam fc2", 2048, 768, 3072),
    ("sam adapter conv", 512, 768, 6912), ("square 4096", 4096, 4096, 4096),
    ("square 8192", 8192, 8192, 8192),
]
EXPERT = ("experts gate|up (E=2, cap 3834, 2556 routed)", 3834, 22016, 4096, 2556)


def run_variant(v):
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import torch
    from medplib_amd import ops
    dev = torch.device("cuda:0")
    print(f"variant {v}")
    for name, M, N, K in SHAPES:
        a = torch.randn(M, K, device=dev).to(torch.bfloat16)
        # weights are COLD in the real model (22 GB stream per step): rotate over > 512 MiB of distinct weight buffers so the
        # 256 MiB Infinity Cache cannot serve them
        nw = max(2, int(600e6 // (N * K * 2)) + 1)
        ws = [(torch.randn(N, K, device=dev) * 0.05).to(torch.bfloat16) for _ in range(nw)]
        out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
        for i in range(3):
            ops.gemm(a, ws[i % nw], out=out)
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 24
        s.record()
        for i in range(n):
            ops.gemm(a, ws[i % nw], out=out)
        e.record(); torch.cuda.synchronize()
        ms = s.elapsed_time(e) / n
        print(f"  {name:28s} {M:5d}x{N:5d}x{K:5d}  {ms * 1e3:8.1f} us  {2.0 * M * N * K / ms / 1e9:7.1f} TF/s", flush=True)
    # batched expert GEMM with device-side row counts (capacity-sized grid)
    name, cap, N, K, routed = EXPERT
    a = torch.randn(2, cap, K, device=dev).to(torch.bfloat16)
    ws = [(torch.randn(2, N, K, device=dev) * 0.05).to(torch.bfloat16) for _ in range(3)]
    out = torch.empty(2, cap, N, dtype=torch.bfloat16, device=dev)
    cnt = torch.tensor([routed, routed], dtype=torch.int32, device=dev)
    for i in range(3):
        ops.gemm_batched(a, ws[i % 3], out, m_dev=cnt)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for i in range(12):
        ops.gemm_batched(a, ws[i % 3], out, m_dev=cnt)
    e.record(); torch.cuda.synchronize()
    ms = s.elapsed_time(e) / 12
    print(f"  {name:28s} {ms * 1e3:8.1f} us  {2.0 * 2 * routed * N * K / ms / 1e9:7.1f} TF/s", flush=True)
    # expert down projection (N = 4096: 2 x 160 tiles -> the tail split-K case of the flat batched decode)
    N, K = 4096, 11008
    a = torch.randn(2, cap, K, device=dev).to(torch.bfloat16)
    ws = [(torch.randn(2, N, K, device=dev) * 0.05).to(torch.bfloat16) for _ in range(3)]
    out = torch.empty(2, cap, N, dtype=torch.bfloat16, device=dev)
    for i in range(3):
        ops.gemm_batched(a, ws[i % 3], out, m_dev=cnt)
    torch.cuda.synchronize()
    s.record()
    for i in range(12):
        ops.gemm_batched(a, ws[i % 3], out, m_dev=cnt)
    e.record(); torch.cuda.synchronize()
    ms = s.elapsed_time(e) / 12
    print(f"  {'experts down (E=2, 2556 routed)':28s} {ms * 1e3:8.1f} us  {2.0 * 2 * routed * N * K / ms / 1e9:7.1f} TF/s", flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--child":
        run_variant(os.environ.get("MP_GEMM_VARIANT", "1"))
    else:
        for v in (sys.argv[1:] or ["1", "2"]):
            var, _, rest = v.partition(":")
            grp, _, abl = rest.partition(":")
            print(f"== variant {var} group_m {grp or 'default'} ablate {abl or 0}", flush=True)
            env = dict(os.environ, MP_GEMM_VARIANT=var)
            if grp:
                env["MP_GEMM_GROUP_M"] = grp
            if abl:
                env["MP_GEMM_ABLATE"] = abl
            subprocess.run([sys.executable, __file__, "--child"], env=env)
