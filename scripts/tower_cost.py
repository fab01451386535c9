"""What the frozen towers cost the stage-III step in the overlapped regime: the same step with the CLIP tower's features and / or the SAM
encoder's embedding of step 1 reused (a MEASUREMENT hook: the wrappers live here, not in the product) — the upper bound of what a faster
tower could give.  python scripts/tower_cost.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench as B
from medplib_amd import engine
from medplib_amd.model.config import MedPLIBConfig
from medplib_amd.model.medplib import MedPLIBForCausalLM

dev = torch.device("cuda:0")
torch.manual_seed(1234)
cfg = MedPLIBConfig.medplib_7b()
model = MedPLIBForCausalLM(cfg, device=dev).train()
model.towers_run_ahead = True
eng, _, _, _ = engine.initialize(model=model, model_parameters=model.trainable_parameters(),
                                 config={"optimizer": {"params": {"lr": 3e-4, "betas": (0.9, 0.95)}}, "gradient_clipping": 1.0})
batch = B.synthetic_batch(cfg, 8, dev, seed=42)
real_enc, real_sam = model._encode_and_plan, model.get_visual_embs
cache = {}
def enc(*a, **k):
    if "clip" not in cache: cache["clip"] = real_enc(*a, **k)
    return cache["clip"]
def sam(*a, **k):
    if "sam" not in cache: cache["sam"] = real_sam(*a, **k)
    return cache["sam"]
def run(tag, n=20, w=5):
    for _ in range(w):
        o = eng(**batch); eng.backward(o); eng.step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        o = eng(**batch); eng.backward(o); eng.step()
    torch.cuda.synchronize()
    print(f"{tag:32s} {(time.perf_counter() - t0) / n * 1e3:7.2f} ms/step", flush=True)
run("all towers computed")
model._encode_and_plan = enc; run("CLIP tower + projector reused")
model._encode_and_plan = real_enc; model.get_visual_embs = sam; run("SAM encoder reused")
model._encode_and_plan = enc; run("both reused (decoder + tail only)")
