import os, sys, torch, json
sys.path.insert(0, os.getcwd())
from medplib_amd.model.config import MedPLIBConfig
from oracle.parity import full_size_parity
dev = torch.device("cuda:0")
torch.set_num_threads(min(32, os.cpu_count()))
for kw in (dict(num_experts=4, top_k_experts=2), dict(num_experts=2, top_k_experts=1, use_residual=True), dict(num_experts=4, top_k_experts=2, use_residual=True)):
    cfg = MedPLIBConfig.medplib_7b(num_hidden_layers=4, vocab_size=4096, seg_token_idx=4000, moe_enable=True, **kw)
    r = full_size_parity(cfg, dev)
    print(kw, {k: (round(v, 5) if isinstance(v, float) else v) for k, v in r.items() if k in ("abs_dloss", "max_abs_dloss_over_10", "hidden_rel_err", "hidden_mean_rel_err", "loss_gpu", "loss_cpu", "capacity", "abs_ddice", "hidden_rel_err_agreeing_rows", "rows_agreeing_in_every_layer", "routing_agreement_per_layer")}, r["mask"]["max_abs_dlogit"], r.get("routing"), flush=True)
