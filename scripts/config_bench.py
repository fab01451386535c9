"""BASELINE.json configs[1] and configs[4] at the true 7B dimensions on one MI355X (run on the GPU box) — the two configurations that
are parity-test cases at tiny dims (tests/test_gpu_model.py), measured here so their cost at scale is on record:
  --config 1   MedPLIB-7B bf16 VQA-only forward (MoE disabled), batch 4: CE-only batch (seg_flag False), no SAM work
  --config 4   MedPLIB-ICL separate mode: 3 in-context (image, mask) pairs + query image per sample, mm_token_compress 576 -> 256,
               MaskTokenEncoder 64 tokens, MoE E=2 top-1, batch 4 (the per-GPU share; expert parallelism needs > 1 rank)
Synthetic inputs as SURVEY 8(d) prescribes.  Prints one JSON line."""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from medplib_amd.model.config import MedPLIBConfig
from medplib_amd.model.medplib import LISAForCausalLM, MedPLIBForCausalLM

ap = argparse.ArgumentParser()
ap.add_argument("--config", type=int, required=True, choices=(1, 4))
ap.add_argument("--steps", type=int, default=10)
ap.add_argument("--warmup", type=int, default=3)
args = ap.parse_args()
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(42)
B = 4


def discs(n, size):
    yy, xx = torch.meshgrid(torch.arange(size), torch.arange(size), indexing="ij")
    out = []
    for _ in range(n):
        cy, cx = (torch.rand(2, generator=g) * size).tolist()
        r = 20 + 80 * torch.rand(1, generator=g).item()
        out.append((((yy - cy) ** 2 + (xx - cx) ** 2) < r * r).float())
    return out


if args.config == 1:
    cfg = MedPLIBConfig.medplib_7b(moe_enable=False)
    model = LISAForCausalLM(cfg, device=dev).eval()
    L, V = 64, cfg.vocab_size
    ids = torch.randint(3, 31999, (B, L), generator=g)
    ids[:, 0] = 1; ids[:, 34], ids[:, 35], ids[:, 36] = V - 2, -200, V - 1; ids[:, 63] = 2
    labels = ids.clone(); labels[:, :56] = -100
    batch = {"images": torch.randn(B, 3, 256, 256, generator=g).to(dev), "images_clip": torch.randn(B, 3, 336, 336, generator=g).to(torch.bfloat16).to(dev),
             "input_ids": ids.numpy(), "labels": labels.numpy(), "attention_mask": torch.ones(B, L, dtype=torch.bool).numpy(),
             "masks_list": [], "label_list": [], "resize_list": [(256, 256)] * B, "valid_mask_bool": [[]] * B, "offset": None,
             "region_masks": [], "inference": False, "seg_flag": False}
    S = L - 1 + cfg.clip_num_patches
    what = "MedPLIB-7B bf16 VQA-only forward (MoE disabled), batch 4, CE loss"
else:
    n_ctx = 3
    cfg = MedPLIBConfig.medplib_7b(mm_token_compress=True, mm_compressed_token_count=256, icl_mask_encoder=True, mask_encoder_token_count=64)
    model = MedPLIBForCausalLM(cfg, device=dev).eval()
    V = cfg.vocab_size
    n_ph = 2 * n_ctx + 1
    L = 8 + 4 * n_ph + 28
    ids = torch.randint(3, 31999, (B, L), generator=g)
    ids[:, 0] = 1
    for k in range(n_ph):
        p = 6 + 4 * k
        ids[:, p - 1], ids[:, p], ids[:, p + 1] = V - 2, -200, V - 1
    ids[:, L - 3] = cfg.seg_token_idx; ids[:, L - 1] = 2
    labels = torch.full((B, L), -100, dtype=torch.int64); labels[:, L - 8:] = ids[:, L - 8:]
    lengths = [[256, 64] * n_ctx + [256] for _ in range(B)]
    batch = {"images": torch.randn(B, 3, 256, 256, generator=g).to(dev),
             "images_clip": [torch.randn(n_ctx + 1, 3, 336, 336, generator=g).to(torch.bfloat16).to(dev) for _ in range(B)],
             "mask_images": [torch.stack(discs(n_ctx, 336)).unsqueeze(1).to(dev) for _ in range(B)],
             "image_token_types": [["image", "mask"] * n_ctx + ["image"] for _ in range(B)], "image_token_lengths": lengths,
             "icl_image_counts": [n_ctx + 1] * B,
             "input_ids": ids.numpy(), "labels": labels.numpy(), "attention_mask": torch.ones(B, L, dtype=torch.bool).numpy(),
             "masks_list": [m.to(dev) for m in discs(B, 336)], "label_list": [torch.empty(336, 336, device="meta") for _ in range(B)],
             "resize_list": [(256, 256)] * B, "valid_mask_bool": [[True]] * B, "offset": None, "region_masks": [],
             "inference": False, "seg_flag": True}
    S = L - n_ph + sum(lengths[0])
    what = ("MedPLIB-ICL separate mode forward: 3 in-context (image, mask) pairs + query, token compressor 576->256, mask encoder 64 tokens, "
            "MoE E=2 top-1, batch 4, CE + mask losses")

with torch.no_grad():
    for _ in range(args.warmup):
        out = model(**batch)
    model.sync_side_streams(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = model(**batch)
    model.sync_side_streams(); torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.steps
d, ff, nl = cfg.hidden_size, cfg.intermediate_size, cfg.num_hidden_layers
llm_flop = B * S * nl * 2 * (4 * d * d + 3 * d * ff) + B * nl * 4 * S * S * d / 2
n_img = B if args.config == 1 else B * 4
clip_flop = n_img * 0.366e12
print(json.dumps({"config": args.config, "what": what, "batch": B, "seq_len_after_splice": S, "ms_per_forward": round(dt * 1e3, 2),
                  "samples_per_s": round(B / dt, 2), "llm_plus_clip_tflop": round((llm_flop + clip_flop) / 1e12, 2),
                  "tflops": round((llm_flop + clip_flop) / dt / 1e12, 1), "mfma_utilisation": round((llm_flop + clip_flop) / dt / 2.5e15, 4),
                  "loss": float(out["loss"]) if "loss" in out else None, "dtype": "bf16", "data": "synthetic"}))
