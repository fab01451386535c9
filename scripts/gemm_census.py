import os, sys, collections, torch
sys.path.insert(0, os.getcwd())
import bench
from medplib_amd import engine, ops
from medplib_amd.model.config import MedPLIBConfig
from medplib_amd.model.medplib import MedPLIBForCausalLM
dev = torch.device("cuda:0")
cfg = MedPLIBConfig.medplib_7b(num_hidden_layers=2)
model = MedPLIBForCausalLM(cfg, device=dev).train()
eng, _, _, _ = engine.initialize(model=model, model_parameters=model.trainable_parameters(), config={"train_micro_batch_size_per_gpu": 8, "optimizer": {"params": {"lr": 1e-4}}})
batch = bench.synthetic_batch(cfg, 8, dev, 42)
def step():
    out = eng(**batch); eng.backward(out); eng.step()
step(); torch.cuda.synchronize()
real = ops.gemm
seen = collections.Counter()
def wrapped(a, w, bias=None, residual=None, act=ops.ACT_NONE, out_dtype=torch.bfloat16, out=None, alpha=1.0, m_dev=None):
    r = real(a, w, bias=bias, residual=residual, act=act, out_dtype=out_dtype, out=out, alpha=alpha, m_dev=m_dev)
    seen[(ops.gemm_last_kernel(), a.shape[0], w.shape[0], a.shape[1], int(act), bias is not None, residual is not None, str(out_dtype).split(".")[-1])] += 1
    return r
ops.gemm = wrapped
import medplib_amd.model.sam as S, medplib_amd.model.clip as C
step(); torch.cuda.synchronize()
for k, n in sorted(seen.items(), key=lambda kv: (kv[0][0], -kv[1])):
    print("kernel %d  M=%d N=%d K=%d act=%d bias=%s res=%s out=%s  x%d" % (k + (n,)))
