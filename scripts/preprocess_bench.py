"""Image-preprocessing kernels vs the HBM roofline, with PIL on the host beside them (run on the GPU box).
Algorithmic bytes of one image: read H*W*3 once, write the two resized images and the two normalised CHW tensors."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from medplib_amd import preprocess as PP

dev = torch.device("cuda:0")
rng = np.random.default_rng(0)
res = []
for h, w in [(512, 512), (1024, 1024), (3000, 4000)]:
    img_np = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    img = torch.from_numpy(img_np).to(dev)

    def run():
        s, _ = PP.preprocess_sam(img)
        c = PP.preprocess_clip(img)
        return s, c
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 50
    s.record()
    for _ in range(n):
        run()
    e.record(); torch.cuda.synchronize()
    us = s.elapsed_time(e) / n * 1e3
    sh, sw = PP.get_preprocess_shape(h, w, 256)
    ch, cw = PP.get_preprocess_shape(h, w, 336)
    alg = 2 * h * w * 3 + (sh * sw + ch * cw) * 3 * 2 + (3 * 256 * 256 + 3 * 336 * 336) * 4     # each pipeline reads the source once
    cpu_us = None
    try:
        from PIL import Image
        t0 = time.perf_counter()
        for _ in range(5):
            im = Image.fromarray(img_np)
            a = np.array(im.resize((sw, sh), Image.BILINEAR)); b = np.array(im.resize((cw, ch), Image.BILINEAR))
        cpu_us = (time.perf_counter() - t0) / 5 * 1e6
    except ImportError:
        pass
    res.append({"image": [h, w], "gpu_us_per_image": round(us, 1), "algorithmic_bytes": alg, "achieved_GBps": round(alg / us / 1e3, 1),
                "frac_of_8TBps": round(alg / us / 1e3 / 8000, 4), "pil_resize_only_cpu_us": None if cpu_us is None else round(cpu_us, 1)})
    print(res[-1], flush=True)
print(json.dumps({"what": "preprocess_sam + preprocess_clip per image (6 launches)", "results": res}))
