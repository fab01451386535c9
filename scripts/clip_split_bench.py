"""CLIP tower on the batch of 8: one chain vs two half-batch chains on two streams (do the small GEMMs' ramp / drain phases overlap?).
python scripts/clip_split_bench.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from medplib_amd.model.config import MedPLIBConfig
from medplib_amd.model.medplib import MedPLIBForCausalLM
dev = torch.device("cuda:0")
cfg = MedPLIBConfig.medplib_7b(num_hidden_layers=1)
model = MedPLIBForCausalLM(cfg, device=dev).eval()
g = torch.Generator().manual_seed(0)
img = torch.randn(8, 3, 336, 336, generator=g).to(dev).to(torch.bfloat16)
tower = model.model.vision_tower
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()

def whole():
    return tower.encode_images(img)

def split(parts):
    cur = torch.cuda.current_stream()
    outs = []
    n = img.shape[0] // parts
    streams = [s1, s2][:parts] if parts <= 2 else [torch.cuda.Stream() for _ in range(parts)]
    for i, st in enumerate(streams):
        st.wait_stream(cur)
        with torch.cuda.stream(st):
            outs.append(tower.encode_images(img[i * n:(i + 1) * n]))
    for st in streams:
        cur.wait_stream(st)
    return outs

def timed(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n

with torch.no_grad():
    print(f"one chain, batch 8: {timed(whole):.3f} ms")
    print(f"two chains of 4 on two streams: {timed(lambda: split(2)):.3f} ms")
    print(f"one chain, batch 4 alone: {timed(lambda: tower.encode_images(img[:4])):.3f} ms")
    print(f"one chain, batch 8: {timed(whole):.3f} ms")
