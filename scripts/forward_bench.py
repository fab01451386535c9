"""BASELINE config 3: MedPLIB-7B-MoE bf16 pixel-grounding FORWARD (CLIP + splice + 32-layer MoE Llama + SAM-Med2D encoder + prompt /
mask decoder + postprocess), batch 8, one MI355X — the configuration the ">= 40 % MFMA utilisation on the 7B-MoE forward" target is
stated on.  Prints one JSON line: ms per forward, samples/s, model TFLOP/s (9.15 TFLOP/sample algorithmic) and its
fraction of the 2.5 PFLOP/s dense bf16 peak.  python scripts/forward_bench.py [--steps K] [--warmup W]"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from medplib_amd.model.config import MedPLIBConfig
from medplib_amd.model.medplib import MedPLIBForCausalLM

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=10)
ap.add_argument("--warmup", type=int, default=3)
ap.add_argument("--batch", type=int, default=8)
args = ap.parse_args()
dev = torch.device("cuda:0")
torch.set_num_threads(8)
cfg = MedPLIBConfig.medplib_7b()
model = MedPLIBForCausalLM(cfg, device=dev).eval()
batch = bench.synthetic_batch(cfg, args.batch, dev, 42)
batch["inference"] = True
with torch.no_grad():
    for _ in range(args.warmup):
        out = model(**batch)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = model(**batch)
    torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / args.steps
tf = bench.FWD_TFLOP_PER_SAMPLE * args.batch / dt
# the language model alone (32 MoE decoder layers + final norm on [B, 639, 4096] embeddings): 8.66 TFLOP / sample minus lm_head's
# share on unsupervised rows is not subtracted — the stack's own algorithmic work is 32 x 2 x 639 x 202.4 M + attention
S = 639
emb = (torch.randn(args.batch, S, cfg.hidden_size, device=dev) * 0.5).to(torch.bfloat16)
llm = model.model.llm
with torch.no_grad():
    for _ in range(args.warmup):
        llm.forward(emb, None)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for _ in range(args.steps):
        llm.forward(emb, None)
    torch.cuda.synchronize()
dt_llm = (time.perf_counter() - t1) / args.steps
d, ff, H, D, Lr = cfg.hidden_size, cfg.intermediate_size, cfg.num_attention_heads, cfg.head_dim, cfg.num_hidden_layers
flop_llm = args.batch * Lr * (2.0 * S * (4 * d * d + 3 * d * ff) + 4.0 * S * S * D * H)          # linears + QK^T / PV (full S^2, SURVEY's convention)
tf_llm = flop_llm / dt_llm / 1e12
print(json.dumps({"metric": "forward samples/sec (BASELINE config 3: 7B-MoE pixel-grounding forward + SAM-Med2D decoder, batch 8)",
                  "value": round(args.batch / dt, 2), "unit": "samples/s", "ms_per_forward": round(dt * 1e3, 2), "steps": args.steps,
                  "warmup": args.warmup, "model_tflops": round(tf, 1), "mfma_peak_tflops": bench.MFMA_BF16_PEAK_TFLOPS,
                  "mfma_utilisation": round(tf / bench.MFMA_BF16_PEAK_TFLOPS, 4),
                  "llm_stack": {"what": "32-layer 7B-MoE decoder stack alone, [8, 639, 4096] embeddings", "ms": round(dt_llm * 1e3, 2),
                                "tflop": round(flop_llm / 1e12, 2), "tflops": round(tf_llm, 1),
                                "mfma_utilisation": round(tf_llm / bench.MFMA_BF16_PEAK_TFLOPS, 4)},
                  "dtype": "bf16", "data": "synthetic",
                  "n_masks": len(out["pred_masks"])}))
