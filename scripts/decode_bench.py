"""evaluate() at the 7B dimensions: prefill + greedy single-token decode steps with the KV cache.  Prints ms per
decode step and the HBM rate of the weight stream (every decode step reads all LLM weights once: GEMV-bound).
python scripts/decode_bench.py [--new 32] [--dense]"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from medplib_amd.model.config import MedPLIBConfig
from medplib_amd.model.medplib import LISAForCausalLM, MedPLIBForCausalLM

ap = argparse.ArgumentParser()
ap.add_argument("--new", type=int, default=32)
ap.add_argument("--dense", action="store_true")
ap.add_argument("--no-fuse-routing", action="store_true", help="A/B: separate rmsnorm / gate / route launches in the decode step")
args = ap.parse_args()
dev = torch.device("cuda:0")
cfg = MedPLIBConfig.medplib_7b(moe_enable=not args.dense)
model = (LISAForCausalLM if args.dense else MedPLIBForCausalLM)(cfg, device=dev).eval()
if args.no_fuse_routing:
    model.model.llm.fuse_decode_routing = False
g = torch.Generator().manual_seed(0)
L, V = 64, cfg.vocab_size
ids = torch.randint(3, 31999, (1, L), generator=g)
ids[0, 0] = 1; ids[0, 34], ids[0, 35], ids[0, 36] = V - 2, -200, V - 1
images_clip = torch.randn(1, 3, 336, 336, generator=g).to(torch.bfloat16).to(dev)
images = torch.randn(1, 3, 256, 256, generator=g).to(dev)
res = {}
# one untimed call first: code-object loading and other first-use costs belong to neither of the two lengths the slope is taken between
model.evaluate(images_clip, images, ids.numpy(), [(256, 256)], [(336, 336)], max_new_tokens=8, eos_token_id=-1)
for n_new in (args.new, 4 * args.new):      # slope between two lengths: prefill and the one-off graph capture cancel out
    best = float("inf")
    for _ in range(3):                      # the fastest of three calls per length: a one-off hiccup in either length tilts the slope
        torch.cuda.synchronize(); t0 = time.perf_counter()
        out_ids, masks = model.evaluate(images_clip, images, ids.numpy(), [(256, 256)], [(336, 336)], max_new_tokens=n_new, eos_token_id=-1)
        torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    res[n_new] = best
steps = 3 * args.new
ms = (res[4 * args.new] - res[args.new]) / steps * 1e3
d, ff = cfg.hidden_size, cfg.intermediate_size
per_layer = 4 * d * d + 3 * d * ff                       # one expert's MLP is read per token with top-1 routing
wbytes = (cfg.num_hidden_layers * per_layer + V * d) * 2
print(json.dumps({"metric": "decode ms/token (evaluate(), batch 1, KV cache)", "value": round(ms, 3), "total_ms_for_%d_tokens" % args.new: round(res[args.new] * 1e3, 1),
                  "weight_bytes_per_token": wbytes, "weight_stream_GBps": round(wbytes / ms / 1e6, 1), "frac_of_8TBps": round(wbytes / (ms * 1e-3) / 8e12, 4),
                  "moe": not args.dense, "generated": int(out_ids.shape[1] - L)}))
