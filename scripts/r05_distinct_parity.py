"""Round-4 review, parity item (c): ONE full-depth parity run with DISTINCT seeded weights in every decoder layer (the standing full-size
checks alias one layer's weights over all layers on both sides, so a per-layer weight-indexing error at depth > 8 could not show there).
MedPLIB-7B-MoE, 32 layers, E = 2 top-1, B = 1 (S = 639), gate sampling off: the whole model_forward on the CPU oracle (fp32) and on the HIP
path from the same 21.6 GB of bf16-valued decoder weights -> the same bounds as every other full-size run (oracle/parity.check_full_size).
Usage (GPU box): python scripts/r05_distinct_parity.py [layers] [dense] > gpurun_out/r05_distinct_parity.json
(`dense`: the same run on the dense decoder — LISA / the base of every LoRA configuration; round 6)"""
import json
import os
import sys
import time

ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from medplib_amd.model.config import MedPLIBConfig  # noqa: E402
from oracle import parity  # noqa: E402


def main():
    layers = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    moe = not (len(sys.argv) > 2 and sys.argv[2] == "dense")
    cfg = MedPLIBConfig.medplib_7b(num_hidden_layers=layers, moe_enable=moe)
    t0 = time.time()
    r = parity.full_size_parity(cfg, torch.device("cuda:0"), B=1, distinct_weights=True, cpu_threads=os.cpu_count())
    bad = parity.check_full_size(r, layers, moe)
    r["violated_bounds"] = bad
    r["wall_seconds"] = round(time.time() - t0, 1)
    r["host_cores"] = os.cpu_count()
    print(json.dumps(r))
    print("[distinct-weights parity]", "OK" if not bad else "VIOLATED: " + "; ".join(bad), file=sys.stderr)
    sys.exit(3 if bad else 0)


if __name__ == "__main__":
    main()
