"""One frozen tower alone under a kernel trace: TOWER=sam|clip python scripts/r06_tower_trace.py [n] — n forwards at batch 8, true dims,
nothing beside them, so the trace's durations are the kernels' own (the step's table shows them stretched by the decoder's tiles)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from medplib_amd.model.config import MedPLIBConfig
from medplib_amd.model.medplib import MedPLIBForCausalLM

dev = torch.device("cuda:0")
cfg = MedPLIBConfig.medplib_7b(num_hidden_layers=1)
model = MedPLIBForCausalLM(cfg, device=dev).eval()
g = torch.Generator().manual_seed(0)
which = os.environ.get("TOWER", "sam")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10
x = (torch.randn(8, 3, 256, 256, generator=g).to(dev) if which == "sam" else torch.randn(8, 3, 336, 336, generator=g).to(dev).to(torch.bfloat16))
fn = (lambda: model.get_visual_embs(x)) if which == "sam" else (lambda: model.model.vision_tower.encode_images(x))
with torch.no_grad():
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record(); torch.cuda.synchronize()
print(which, "ms per forward", s.elapsed_time(e) / n, flush=True)
