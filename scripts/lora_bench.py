"""LoRA training step at the true 7B dimensions: dense Llama-7B + adapters (r = 8, alpha = 16) on
gate/up/down_proj — scripts/train_stage3.sh's targets — with the mask decoder and text_hidden_fcs trainable as in stage III,
per-GPU batch 8, synthetic inputs of bench.py.  One JSON line (run on the GPU box)."""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from medplib_amd import engine
from medplib_amd.model.config import MedPLIBConfig
from medplib_amd.model.medplib import LISAForCausalLM, MedPLIBForCausalLM

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=5)
ap.add_argument("--warmup", type=int, default=2)
ap.add_argument("--batch", type=int, default=8)
ap.add_argument("--lora_r", type=int, default=8)
ap.add_argument("--lora_dropout", type=float, default=0.05)
ap.add_argument("--targets", type=str, default="gate_proj,up_proj,down_proj")
ap.add_argument("--sft", type=str, default="mask_decoder,text_hidden_fcs", help="--sft_modules (lm_head, embed_tokens, input_layernorm, post_attention_layernorm, mm_projector, wg)")
ap.add_argument("--moe", action="store_true", help="MoE decoder (E = 2, top-1): per-expert adapters + trainable gate (stage IV / ICL scripts)")
args = ap.parse_args()
dev = torch.device("cuda:0")
torch.manual_seed(1234)
cfg = MedPLIBConfig.medplib_7b(moe_enable=args.moe)
model = (MedPLIBForCausalLM if args.moe else LISAForCausalLM)(cfg, device=dev).train()
lora = model.enable_lora(lora_r=args.lora_r, lora_alpha=16, lora_dropout=args.lora_dropout, lora_target_modules=args.targets, sft_modules=args.sft)
for n, p in zip(lora.names, lora.params):                  # B = 0 at initialisation would make half the gradients trivially zero
    if "lora_B" in n:
        p.data.normal_(0, 0.01)
eng, _, _, _ = engine.initialize(model=model, model_parameters=model.trainable_parameters(),
                                 config={"train_micro_batch_size_per_gpu": args.batch, "optimizer": {"params": {"lr": 1e-4, "betas": (0.9, 0.95)}},
                                         "gradient_clipping": 1.0})
batch = bench.synthetic_batch(cfg, args.batch, dev, seed=42)


def step():
    out = eng(**batch)
    eng.backward(out["loss"])
    eng.step()
    return out


for _ in range(args.warmup):
    out = step()
torch.cuda.synchronize()
l0 = float(out["loss"].detach())
torch.cuda.reset_peak_memory_stats()
t0 = time.perf_counter()
for _ in range(args.steps):
    out = step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / args.steps
S, d, ff, nl = 639, cfg.hidden_size, cfg.intermediate_size, cfg.num_hidden_layers
T = args.batch * S
gemm_flop = 2 * T * nl * (4 * d * d + 3 * d * ff) * 2                   # forward + dgrad of the frozen projections
attn_flop = args.batch * nl * cfg.num_attention_heads * S * S * cfg.head_dim * (4 + 14) / 2      # causal: fwd 2 + bwd 7 matmuls
print(json.dumps({"what": "%s 7B + LoRA (%s, r=%d, dropout %.2f) training step, batch %d" % ("MoE (E=2, top-1, wg trainable)" if args.moe else "dense", args.targets, args.lora_r, args.lora_dropout, args.batch),
                  "ms_per_step": round(dt * 1e3, 1), "samples_per_s": round(args.batch / dt, 2),
                  "decoder_tflop_per_step": round((gemm_flop + attn_flop) / 1e12, 1), "decoder_tflops": round((gemm_flop + attn_flop) / dt / 1e12, 1),
                  "sft_modules": args.sft, "trainable_params": eng.optimizer.numel, "peak_mem_GB": round(torch.cuda.max_memory_allocated() / 2 ** 30, 1),
                  "loss_after_warmup": l0, "loss_last": repr(float(out["loss"].detach()))}))
