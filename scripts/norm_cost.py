"""Upper bound of what folding the decoder's input RMSNorm into the qkv GEMM could give: the stage-III step with ops.rmsnorm on the
[5112, 4096] residual stream SKIPPED (the qkv GEMM reads the residual stream itself: wrong numbers, right timing).  A MEASUREMENT hook,
not a product path.  python scripts/norm_cost.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench as B
from medplib_amd import engine, ops
from medplib_amd.model import llama
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
real = ops.rmsnorm
cache = {}
def fake(x, w, eps, *a, **k):
    if x.shape[0] == 5112:
        # rotate over 40 stale outputs (1.7 GB: colder than the 256 MB Infinity Cache, like a freshly written residual stream; a single cached
        # buffer would sit in the caches, and feeding the un-normalised stream overflows and lets the chip clock up on NaNs)
        if len(cache) < 40:
            cache[len(cache)] = real(x, w, eps, *a, **k).clone()
            return cache[len(cache) - 1]
        fake.i = (getattr(fake, "i", 0) + 1) % 40
        return cache[fake.i]
    return real(x, w, eps, *a, **k)
def run(tag, n=20, wu=5):
    for _ in range(wu):
        o = eng(**batch); eng.backward(o); eng.step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        o = eng(**batch); eng.backward(o); eng.step()
    torch.cuda.synchronize()
    print(f"{tag:40s} {(time.perf_counter() - t0) / n * 1e3:7.2f} ms/step", flush=True)
run("as is")
ops.rmsnorm = fake; llama.ops.rmsnorm = fake
run("input RMSNorm launches skipped (33/step)")
ops.rmsnorm = real; llama.ops.rmsnorm = real
run("as is (again)")
real_fill = ops.moe_fill_dropped
ops.moe_fill_dropped = lambda *a, **k: None
run("moe_fill_dropped launches skipped (32/step)")
ops.rmsnorm = fake; llama.ops.rmsnorm = fake
run("both skipped")
ops.rmsnorm = real; llama.ops.rmsnorm = real; ops.moe_fill_dropped = real_fill
run("as is (third)")
