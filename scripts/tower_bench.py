"""The two frozen towers alone, batch 8 at true dimensions: CLIP ViT-L/14-336 (+ mm_projector) is on the decoder's critical path every
step, the SAM-Med2D encoder runs beside it on a side stream.  python scripts/tower_bench.py"""
import os
import sys
import json

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from medplib_amd import ops
from medplib_amd.model.config import MedPLIBConfig
from medplib_amd.model.medplib import MedPLIBForCausalLM

dev = torch.device("cuda:0")
cfg = MedPLIBConfig.medplib_7b(num_hidden_layers=1)
model = MedPLIBForCausalLM(cfg, device=dev).eval()
g = torch.Generator().manual_seed(0)
clip_img = torch.randn(8, 3, 336, 336, generator=g).to(dev).to(torch.bfloat16)
sam_img = torch.randn(8, 3, 256, 256, generator=g).to(dev)


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n


res = {}
with torch.no_grad():
    res["clip_tower_plus_projector_ms"] = round(timed(lambda: model.model.vision_tower.encode_images(clip_img)), 3)
    res["sam_encoder_ms"] = round(timed(lambda: model.get_visual_embs(sam_img)), 3)
    timer = ops.KernelTimer(sample_every=1)
    ops.GEMM_TIMER = timer
    model.model.vision_tower.encode_images(clip_img)
    torch.cuda.synchronize()
    ops.GEMM_TIMER = None
    by = {}
    for w, s0, e0, kern in timer.records:
        by.setdefault((kern, round(w / 1e9, 1)), []).append(s0.elapsed_time(e0) * 1e3)
    res["clip_gemms"] = {f"gemm{k} {gf} GFLOP": {"launches": len(v), "avg_us": round(sum(v) / len(v), 1), "TF/s": round(gf / (sum(v) / len(v)) * 1e3, 1)}
                         for (k, gf), v in sorted(by.items())}
    res["clip_gemm_ms_total"] = round(sum(sum(v) for v in by.values()) / 1e3, 3)
print(json.dumps(res, indent=1))
