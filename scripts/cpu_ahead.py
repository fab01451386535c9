"""How far does the host run ahead of the GPU in the training step?  Prints the host-side duration of each un-synchronised step()
call next to the synchronised step time, and a cProfile of one step's host work."""
import cProfile
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from medplib_amd import engine
from medplib_amd.model.config import MedPLIBConfig
from medplib_amd.model.medplib import MedPLIBForCausalLM

device = torch.device("cuda:0")
cfg = MedPLIBConfig.medplib_7b()
model = MedPLIBForCausalLM(cfg, device=device).train()
ds = {"train_micro_batch_size_per_gpu": 8, "gradient_accumulation_steps": 1,
      "optimizer": {"type": "AdamW", "params": {"lr": 3e-4, "weight_decay": 0.0, "betas": (0.9, 0.95)}}, "gradient_clipping": 1.0}
eng, _, _, _ = engine.initialize(model=model, model_parameters=model.trainable_parameters(), config=ds)
batch = bench.synthetic_batch(cfg, 8, device, 42)


def step():
    out = eng(**batch)
    eng.backward(out["loss"])
    eng.step()
    return out


for _ in range(2):
    step()
torch.cuda.synchronize()
host = []
t0 = time.perf_counter()
for _ in range(5):
    a = time.perf_counter()
    step()
    host.append((time.perf_counter() - a) * 1e3)
torch.cuda.synchronize()
print("host ms per step() call:", [round(h, 1) for h in host], " wall ms/step:", round((time.perf_counter() - t0) * 1e3 / 5, 1))
# phase split on the host side with a sync after each phase (where the host time goes)
pr = cProfile.Profile()
pr.enable()
step()
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
