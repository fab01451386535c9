"""Main-stream occupancy of the stage-III training step measured with events (no profiler): per step, the time the MAIN stream spends between the
first and the last thing the step enqueues on it, next to the wall time per step.  python scripts/step_events.py"""
import os, sys, time
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
    out = eng(**batch); eng.backward(out); eng.step(); return out
for _ in range(3): step()
torch.cuda.synchronize()
N = 8
ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(N)]
t0 = time.perf_counter()
for i in range(N):
    ev[i][0].record(); step(); ev[i][1].record()
torch.cuda.synchronize()
wall = (time.perf_counter() - t0) * 1e3 / N
print(f"wall {wall:.2f} ms/step")
print("main stream, start -> end of a step's own work (ms):", [round(a.elapsed_time(b), 2) for a, b in ev])
print("main stream, end of step i -> start of step i+1 (ms):", [round(ev[i][1].elapsed_time(ev[i + 1][0]), 3) for i in range(N - 1)])
print("main stream, start i -> start i+1 (ms):", [round(ev[i][0].elapsed_time(ev[i + 1][0]), 2) for i in range(N - 1)])
# host side: how long the host takes to ISSUE a step (the step() call returns when everything is enqueued), against the wall time
torch.cuda.synchronize()
host = []
t_all = time.perf_counter()
for i in range(N):
    t1 = time.perf_counter(); step(); host.append((time.perf_counter() - t1) * 1e3)
torch.cuda.synchronize()
print(f"host issue time per step (ms): {[round(h, 1) for h in host]}; wall {(time.perf_counter() - t_all) * 1e3 / N:.2f} ms/step; towers_run_ahead={getattr(model, 'towers_run_ahead', None)}")
