"""Full-size parity at prompt lengths other than the benchmark's 64 (S = 639): odd sequence lengths, ragged right padding (key padding
mask), several samples — 2 MoE layers at the 7B dims against the oracle.  python scripts/seq_len_parity.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from medplib_amd.model.config import MedPLIBConfig
from oracle.parity import full_size_parity
dev = torch.device("cuda:0")
torch.set_num_threads(min(32, os.cpu_count()))
keys = ("seq_len", "tokens", "max_abs_dloss_over_10", "hidden_mean_rel_err", "hidden_rel_err_agreeing_rows", "routing_agreement_min", "abs_ddice")
for (L, B, ragged) in [(17, 1, False), (200, 1, False), (333, 2, False), (64, 3, True), (129, 3, True)]:
    cfg = MedPLIBConfig.medplib_7b(num_hidden_layers=2, vocab_size=4096, seg_token_idx=4000, moe_enable=True)
    r = full_size_parity(cfg, dev, prompt_len=L, B=B, ragged=ragged)
    print(f"L={L} B={B} ragged={ragged}", {k: (round(r[k], 5) if isinstance(r[k], float) else r[k]) for k in keys}, "dlogit", round(r["mask"]["max_abs_dlogit"], 4), flush=True)
    assert r["max_abs_dloss_over_10"] < 5e-2 and r["hidden_mean_rel_err"] < 2 ** -6 and r["routing_agreement_min"] >= 0.97, r
