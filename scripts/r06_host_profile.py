"""Where the HOST time of the LoRA stage-III step goes (round-5 review, weak 4: ~2300 launches per step at 36-45 us of Python + ctypes each):
cProfile over a few steps of scripts/lora_bench.py's step — tottime by function, and the pure issue time of a step (no synchronisation inside).
python scripts/r06_host_profile.py [steps] > gpurun_out/r06_host_profile.txt"""
import cProfile, io, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from medplib_amd import engine
from medplib_amd.model.config import MedPLIBConfig
from medplib_amd.model.medplib import LISAForCausalLM

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 4
dev = torch.device("cuda:0")
torch.manual_seed(1234)
cfg = MedPLIBConfig.medplib_7b(moe_enable=False)
model = LISAForCausalLM(cfg, device=dev).train()
lora = model.enable_lora(lora_r=8, lora_alpha=16, lora_dropout=0.05, lora_target_modules="gate_proj,up_proj,down_proj", sft_modules="mask_decoder,text_hidden_fcs")
for n, p in zip(lora.names, lora.params):
    if "lora_B" in n:
        p.data.normal_(0, 0.01)
eng, _, _, _ = engine.initialize(model=model, model_parameters=model.trainable_parameters(),
                                 config={"train_micro_batch_size_per_gpu": 8, "optimizer": {"params": {"lr": 1e-4, "betas": (0.9, 0.95)}}, "gradient_clipping": 1.0})
batch = bench.synthetic_batch(cfg, 8, dev, seed=42)


def step():
    out = eng(**batch); eng.backward(out["loss"]); eng.step()


for _ in range(3):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    step()
t_issue = (time.perf_counter() - t0) / steps
torch.cuda.synchronize()
t_all = (time.perf_counter() - t0) / steps
print(f"host issue {t_issue * 1e3:.1f} ms/step, step {t_all * 1e3:.1f} ms")
pr = cProfile.Profile()
pr.enable()
for _ in range(steps):
    step()
pr.disable()
torch.cuda.synchronize()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(45)
print(s.getvalue()[:9000])
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(40)
print(s.getvalue()[:8000])
