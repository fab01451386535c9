"""Times pieces of the CPU oracle on the host (used to size bench.py's cpu_baseline sample)."""
import os, sys, time, copy
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from medplib_amd.model.config import MedPLIBConfig
from oracle import model as OM, llm as OL

cfg = MedPLIBConfig.medplib_7b()
print("cores", os.cpu_count(), flush=True)
for nt in (int(a) for a in sys.argv[1:] or ["64"]):
    torch.set_num_threads(nt)
    cfg1 = copy.deepcopy(cfg); cfg1.num_hidden_layers = 1
    t0 = time.time(); W = OM.init_hf_weights(cfg1, seed=0); t_init = time.time() - t0
    emb = torch.randn(1, 639, 4096) * 0.5
    for _ in range(2):
        t0 = time.time(); h, aux = OL.llama_forward(emb, None, W, cfg1); t_layer = time.time() - t0
    img = torch.randn(1, 3, 336, 336)
    t0 = time.time(); f = OL.clip_features(img, W, cfg); t_clip = time.time() - t0
    print(f"threads {nt}: init {t_init:.1f}s, 1 llama layer+norm {t_layer:.2f}s, clip {t_clip:.2f}s", flush=True)
