#!/usr/bin/env python
"""bench.py — MedPLIB hot path on MI355X: training samples/s for "336x336 image + 64-token prompt".

Workload (BASELINE.json configs[3], the config the train-samples/s metric is quoted on): MedPLIB-7B-MoE (Llama-7B dims,
E=2 top-1 experts in all 32 layers, capacity_factor 1.5), CLIP ViT-L/14-336, SAM-Med2D ViT-B@256; stage-III training step
with LoRA off (trainable: mask decoder + text_hidden_fcs, 21.9 M parameters), CE + BCE + Dice + Focal(+IoU) losses,
per-GPU micro-batch 8, data parallel (weak scaling), bf16 trunk / fp32 trainable tail, synthetic data, seeded random
weights at true dimensions.

One step = forward of the whole path + backward through the trainable tail + gradient all-reduce + AdamW step.
Launch:  python bench.py --gpus N --steps K --warmup W          (N > 1 without a launcher: re-executes itself under
                                                                 torch.distributed.run with N ranks on 127.0.0.1)
         python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W
Rank 0 prints ONE JSON line.  The line never claims a GPU count other than the one asked for: `--gpus N` with WORLD_SIZE != N is
refused, and `rccl_ranks` is what the RCCL communicator itself reports (ncclCommCount through the C ABI's mp_comm_count).
`--dry` runs the launch / rendezvous / timing / one-line protocol on CPU ranks over gloo with a stand-in gradient bucket (no GPU,
no model): what tests/test_bench_launch.py drives."""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

MFMA_BF16_PEAK_TFLOPS = 2500.0     # dense bf16 MFMA peak, /opt/skills/guides/MI355X_MICROARCH.md
FWD_TFLOP_PER_SAMPLE = 9.15


def pruned_tflop_per_step(model, cfg, lora):
    """TFLOP per step the last decoder layer's MLP does NOT execute because nothing reads those rows (medplib.model_forward: the MLP of
    the last layer runs on the supervised + <SEG> rows; MP_PRUNE_LAST_MLP=0 turns that off) — taken out of `model_tflops_per_gpu`, which
    therefore counts executed algorithmic work.  -> (TFLOP, "n of T" text or None)."""
    nr = getattr(model, "last_pruned", None)
    if nr is None:
        return 0.0, None
    n, T = nr
    per_row = 6.0 * cfg.hidden_size * cfg.intermediate_size            # gate, up, down: 2 flop per multiply-add
    return (T - n) * per_row * (2 if lora else 1) / 1e12, f"{n} of {T}"     # with adapters also the three input-gradient GEMMs


def synthetic_batch(cfg, B, device, seed):
    """SURVEY §8(d) synthetic inputs: N(0,1) images, 64-token prompt with one <image> placeholder at position 35 bracketed
    by <im_start>/<im_end>, <SEG> at 61, EOS at 63, labels supervised from position 56, one binary disc mask per sample."""
    g = torch.Generator().manual_seed(seed)
    L, V = 64, cfg.vocab_size
    ids = torch.randint(3, 31999, (B, L), generator=g)
    ids[:, 0] = 1
    ids[:, 34], ids[:, 35], ids[:, 36] = V - 2, -200, V - 1
    ids[:, 61] = cfg.seg_token_idx
    ids[:, 63] = 2
    labels = ids.clone()
    labels[:, :56] = -100
    att = torch.ones(B, L, dtype=torch.bool)
    H = W = 336
    yy, xx = torch.meshgrid(torch.arange(H), torch.arange(W), indexing="ij")
    masks = []
    for _ in range(B):
        cy, cx = (torch.rand(2, generator=g) * 336).tolist()
        r = 20 + 80 * torch.rand(1, generator=g).item()
        masks.append((((yy - cy) ** 2 + (xx - cx) ** 2) < r * r).float().to(device))
    return {
        "images": torch.randn(B, 3, 256, 256, generator=g).to(device),
        "images_clip": torch.randn(B, 3, 336, 336, generator=g).to(torch.bfloat16).to(device),
        "input_ids": ids.numpy(), "labels": labels.numpy(), "attention_mask": att.numpy(),     # index tensors stay on the host
        "masks_list": masks, "label_list": [torch.empty(H, W, device="meta") for _ in range(B)],
        "resize_list": [(256, 256)] * B, "valid_mask_bool": [[True]] * B, "offset": None, "region_masks": [],
        "inference": False, "seg_flag": True,
    }


def synthetic_icl_batch(cfg, B, device, seed, n_ctx=3):
    """BASELINE configs[4] inputs: per sample 3 in-context (image, mask) pairs + the query image (ICL separate mode), every image behind
    its own <image> placeholder, image tokens compressed 576 -> 256, every in-context mask 64 mask-encoder tokens; S = 1257 after the splice."""
    g = torch.Generator().manual_seed(seed)
    V = cfg.vocab_size

    def discs(n, size):
        yy, xx = torch.meshgrid(torch.arange(size), torch.arange(size), indexing="ij")
        out = []
        for _ in range(n):
            cy, cx = (torch.rand(2, generator=g) * size).tolist()
            r = 20 + 80 * torch.rand(1, generator=g).item()
            out.append((((yy - cy) ** 2 + (xx - cx) ** 2) < r * r).float())
        return out
    n_ph = 2 * n_ctx + 1
    L = 8 + 4 * n_ph + 28
    ids = torch.randint(3, 31999, (B, L), generator=g)
    ids[:, 0] = 1
    for k in range(n_ph):
        p = 6 + 4 * k
        ids[:, p - 1], ids[:, p], ids[:, p + 1] = V - 2, -200, V - 1
    ids[:, L - 3] = cfg.seg_token_idx; ids[:, L - 1] = 2
    labels = torch.full((B, L), -100, dtype=torch.int64); labels[:, L - 8:] = ids[:, L - 8:]
    lengths = [[256, 64] * n_ctx + [256] for _ in range(B)]
    batch = {"images": torch.randn(B, 3, 256, 256, generator=g).to(device),
             "images_clip": [torch.randn(n_ctx + 1, 3, 336, 336, generator=g).to(torch.bfloat16).to(device) for _ in range(B)],
             "mask_images": [torch.stack(discs(n_ctx, 336)).unsqueeze(1).to(device) for _ in range(B)],
             "image_token_types": [["image", "mask"] * n_ctx + ["image"] for _ in range(B)], "image_token_lengths": lengths,
             "icl_image_counts": [n_ctx + 1] * B,
             "input_ids": ids.numpy(), "labels": labels.numpy(), "attention_mask": torch.ones(B, L, dtype=torch.bool).numpy(),
             "masks_list": [m.to(device) for m in discs(B, 336)], "label_list": [torch.empty(336, 336, device="meta") for _ in range(B)],
             "resize_list": [(256, 256)] * B, "valid_mask_bool": [[True]] * B, "offset": None, "region_masks": [],
             "inference": False, "seg_flag": True}
    return batch, L - n_ph + sum(lengths[0])


def icl_config(layers):
    from medplib_amd.model.config import MedPLIBConfig
    return MedPLIBConfig.medplib_7b(num_hidden_layers=layers, mm_token_compress=True, mm_compressed_token_count=256, icl_mask_encoder=True,
                                    mask_encoder_token_count=64)


def forward_rate(model, batch, B, S, n_img, steps=6, warmup=2):
    """ms per forward of the whole path (no_grad, losses computed) and its MFMA fraction from the LLM + CLIP flop."""
    cfg = model.config
    with torch.no_grad():
        for _ in range(warmup):
            out = model(**batch)
        model.sync_side_streams(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            out = model(**batch)
        model.sync_side_streams(); torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
    d, ff, nl = cfg.hidden_size, cfg.intermediate_size, cfg.num_hidden_layers
    flop = B * S * nl * 2 * (4 * d * d + 3 * d * ff) + B * nl * 4 * S * S * d / 2 + n_img * 0.366e12
    return {"batch": B, "seq_len_after_splice": S, "ms_per_forward": round(dt * 1e3, 2), "samples_per_s": round(B / dt, 2),
            "tflops": round(flop / dt / 1e12, 1), "mfma_frac": round(flop / dt / (MFMA_BF16_PEAK_TFLOPS * 1e12), 4),
            "loss": float(out["loss"]) if "loss" in out else None}


def vqa_batch(cfg, B, device, seed):
    """BASELINE configs[1]: VQA-only (CE-only batch: seg_flag False, no masks), batch 4."""
    g = torch.Generator().manual_seed(seed)
    L, V = 64, cfg.vocab_size
    ids = torch.randint(3, 31999, (B, L), generator=g)
    ids[:, 0] = 1; ids[:, 34], ids[:, 35], ids[:, 36] = V - 2, -200, V - 1; ids[:, 63] = 2
    labels = ids.clone(); labels[:, :56] = -100
    return {"images": torch.randn(B, 3, 256, 256, generator=g).to(device),
            "images_clip": torch.randn(B, 3, 336, 336, generator=g).to(torch.bfloat16).to(device),
            "input_ids": ids.numpy(), "labels": labels.numpy(), "attention_mask": torch.ones(B, L, dtype=torch.bool).numpy(),
            "masks_list": [], "label_list": [], "resize_list": [(256, 256)] * B, "valid_mask_bool": [[]] * B, "offset": None,
            "region_masks": [], "inference": False, "seg_flag": False}


def decode_rate(model, device, new=32):
    """evaluate() at batch 1 (model/eval/vqa_infer.py:528-540 -> MedPLIB.py:574-680): ms per decode step as the slope between two lengths
    (prefill and the one-off graph capture cancel), the fastest of three calls per length after one untimed call; every step streams all
    decoder weights once (one expert's MLP per layer with top-1 routing) + lm_head: the HBM fraction of that stream."""
    cfg = model.config
    was = model.training
    model.eval()
    g = torch.Generator().manual_seed(0)
    L, V = 64, cfg.vocab_size
    ids = torch.randint(3, 31999, (1, L), generator=g)
    ids[0, 0] = 1; ids[0, 34], ids[0, 35], ids[0, 36] = V - 2, -200, V - 1
    images_clip = torch.randn(1, 3, 336, 336, generator=g).to(torch.bfloat16).to(device)
    images = torch.randn(1, 3, 256, 256, generator=g).to(device)
    model.evaluate(images_clip, images, ids.numpy(), [(256, 256)], [(336, 336)], max_new_tokens=8, eos_token_id=-1)
    res = {}
    for n_new in (new, 4 * new):
        best = float("inf")
        for _ in range(3):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            model.evaluate(images_clip, images, ids.numpy(), [(256, 256)], [(336, 336)], max_new_tokens=n_new, eos_token_id=-1)
            torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
        res[n_new] = best
    model.train(was)
    ms = (res[4 * new] - res[new]) / (3 * new) * 1e3
    d, ff = cfg.hidden_size, cfg.intermediate_size
    wbytes = (cfg.num_hidden_layers * (4 * d * d + 3 * d * ff) + V * d) * 2
    return {"ms_per_token": round(ms, 3), "weight_bytes_per_token": wbytes, "weight_stream_GBps": round(wbytes / ms / 1e6, 1),
            "frac_of_8TBps": round(wbytes / (ms * 1e-3) / 8e12, 4), "moe": bool(cfg.moe_enable)}


KERNEL_SOURCES = ("gemm320_bf16.hip", "gemm256_bf16.hip", "gemm_bf16.hip", "gemm_common.h", "common.h")


def kernel_source_sha():
    """sha256[:16] over the GEMM kernels' sources: what a PMC profile stamps itself with (scripts/bench_pmc.sh) and what this run compares
    it against — the GPU box has no .git, and 'was this measured on THESE kernels' is the question a commit id only approximates."""
    import hashlib
    h = hashlib.sha256()
    for n in KERNEL_SOURCES:
        with open(os.path.join(ROOT, "medplib_amd", "csrc", n), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def live_traffic(pass_timeout=170):
    """HBM-side bytes per launch of the decoder's 320-row GEMM launches, measured IN THIS RUN: two child passes of this same command (2 steps
    after 1) under `rocprofv3 --kernel-trace --pmc FETCH_SIZE` / `... WRITE_SIZE` (separate passes, kernel-trace only, as the microarchitecture
    guide's HBM section prescribes; read bytes = 2 x FETCH_SIZE KiB on gfx950, WRITE_SIZE KiB raw) — PMC counters cannot be read from inside a
    process, a child under the profiler can.  -> {"read_bytes_per_launch", "write_bytes_per_launch", "launches", "seconds"} or {"error": ...};
    None when rocprofv3 is not installed.  The children print their own JSON line into a pipe that is discarded."""
    import csv, glob, shutil, subprocess, tempfile
    exe = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if exe is None:
        return None
    t0 = time.perf_counter()
    avg = {}
    try:
        clk = []                                         # (shader clocks, ns) of every decoder launch of the WRITE_SIZE pass
        for c in ("FETCH_SIZE", "WRITE_SIZE"):
            d = tempfile.mkdtemp(prefix="mp_pmc_", dir="/tmp")
            # GRBM_GUI_ACTIVE rides the WRITE_SIZE pass (the GRBM block's slots are independent of the TCC's, MI355X_MICROARCH.md "rocprofv3 PMC slots"):
            # the launch's duration in shader clocks, summed over the 8 XCDs -> the clock the chip held under the kernel (round-5 review, item 4)
            pmc = [c] + (["GRBM_GUI_ACTIVE"] if c == "WRITE_SIZE" else [])
            cmd = [exe, "--kernel-trace", "--pmc", *pmc, "--output-format", "csv", "-d", d, "--", sys.executable, os.path.abspath(__file__), "--steps", "2",
                   "--warmup", "1", "--no-cpu-baseline", "--no-lora-line", "--no-secondary", "--no-kernel-timer", "--no-live-traffic"]
            env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
            env["TMPDIR"] = "/tmp"
            r = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True, timeout=pass_timeout)
            vals = []
            for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
                for row in csv.DictReader(open(f)):
                    n = row["Kernel_Name"]
                    if row["Counter_Name"] == c and "gemm320" in n and any(f"kernel<{e}>" in n for e in (1, 2, 3)):      # epilogue families 1-3: the decoder's launches
                        vals.append(float(row["Counter_Value"]))
                    if row["Counter_Name"] == "GRBM_GUI_ACTIVE" and "gemm320" in n and any(f"kernel<{e}>" in n for e in (1, 2, 3)):
                        try:
                            clk.append((float(row["Counter_Value"]) / 8.0, float(row["End_Timestamp"]) - float(row["Start_Timestamp"])))
                        except (KeyError, ValueError):
                            pass
            shutil.rmtree(d, ignore_errors=True)
            if r.returncode != 0 or not vals:
                return {"error": f"{c} pass: rc {r.returncode}, {len(vals)} launches; {r.stderr[-300:]}"}
            avg[c] = (sum(vals) / len(vals), len(vals))
    except Exception as e:
        return {"error": f"{type(e).__name__}: {e}"}
    out = {"read_bytes_per_launch": round(2 * 1024 * avg["FETCH_SIZE"][0]), "write_bytes_per_launch": round(1024 * avg["WRITE_SIZE"][0]),
           "launches": avg["FETCH_SIZE"][1], "seconds": round(time.perf_counter() - t0, 1)}
    clk = [(cy, ns) for cy, ns in clk if cy > 0 and ns > 0]
    if clk:
        out["shader_clocks_per_launch"] = round(sum(cy for cy, _ in clk) / len(clk))
        out["effective_clock_ghz"] = round(sum(cy for cy, _ in clk) / sum(ns for _, ns in clk), 3)
    return out


def cpu_baseline(cfg, device, warmup=3, timed=5):
    """The oracle (CPU fp32 port of the reference path) timed on the host cores on a bounded sample of the same workload:
    BASELINE.md section 3's protocol — 3 warm-up + 5 timed iterations, median and min — for its two configurations: (2) the config-4
    training step at B=1 (forward of the whole path + backward through mask decoder / text_hidden_fcs; this is `value`) and (1) the B=1
    forward alone (`forward_b1`).  To bound host memory and initialisation time the 32 decoder layers alias ONE layer's random weights
    (arithmetic and memory traffic per layer are unchanged: a layer's 1.6 GB of fp32 weights do not fit in cache).
    The same leg then loads the SAME weights into a second, un-timed HIP model and compares the two forwards at the benchmark's
    own batch (`parity`, oracle/parity.py): B = 8 (T = 5112 tokens, capacity 3834), all 32 layers, DeepSpeed's Random Token
    Selection ON with identical uniform draws injected on both sides; the seeded gate is unbalanced enough that the fuller expert
    overflows in every layer, so the draws decide which tokens are dropped — per-layer kept / dropped sets, slots and counts,
    the 10 losses, the last hidden state, and the 8 masks at the reference's threshold and at logit 0."""
    from oracle.parity import full_size_parity
    # 32 threads is the fastest setting measured on the GPU box's 2 x EPYC 9575F for the B=1 step (one llama layer: 0.29 s @32,
    # 0.40 s @64, 0.61 s @128 — the oracle's eager ops stop scaling past one CCD group); `cores` reports what was used.
    threads = min(32, os.cpu_count())
    r = full_size_parity(cfg, device, cpu_threads=min(64, os.cpu_count()), time_threads=threads, time_oracle=(warmup, timed),
                         time_forward=(warmup, timed), B=8, rts_seed=77)
    ts = r.pop("oracle_step_seconds")
    tf = r.pop("oracle_forward_b1_seconds")
    t, f = float(np.median(ts)), float(np.median(tf))
    base = {"value": 1.0 / t, "unit": "samples/s", "cores": threads, "kind": "port",
            "sample": f"oracle fp32 training step at B=1 (1 of the 8 per-GPU samples), true dims, median of {timed} after {warmup} warm-up(s), "
                      f"{t:.2f} s/step (min {min(ts):.2f}); decoder-layer weights aliased across the {cfg.num_hidden_layers} layers",
            "protocol": f"BASELINE.md section 3: {warmup} warm-up + {timed} timed, median and min",
            "forward_b1": {"what": "BASELINE.md section 3 config (1): B=1 forward of the whole path (336x336 image + 64-token prompt, SAM-Med2D encoder, "
                                   "<SEG> projection, mask decoder, upsampler, resize to 336x336), same weights", "seconds_median": round(f, 3),
                           "seconds_min": round(min(tf), 3), "samples_per_s": round(1.0 / f, 4)},
            "step_seconds_median": round(t, 3), "step_seconds_min": round(min(ts), 3)}
    return base, r


def upsampler_roofline(device):
    """The HBM-target kernel of BASELINE.json's north_star (>= 60 % of HBM bandwidth on the SAM mask-decoder upsampler at batch 8):
    `mp_mask_upsample_fused_bf16` timed with HIP events inside this run at batch 8 for the model's real geometry (256-px SAM:
    16 x 16 tokens, 3.29 MB algorithmic) and the 1024-px SAM geometry (64 x 64 tokens, 50.5 MB) — SURVEY §8d.  Algorithmic bytes =
    tokens x 256 ch x 2 B in + packed weights + tokens x 16 px x 32 ch x 2 B out.  Inputs rotate over more buffers than the 256 MiB
    Infinity Cache holds at the large geometry."""
    from medplib_amd import ops
    g = torch.Generator(device=device).manual_seed(0)
    w1 = torch.randn(256, 64, 2, 2, device=device, generator=g) * 0.06
    w2 = torch.randn(64, 32, 2, 2, device=device, generator=g) * 0.12
    w1p, w2p = ops.pack_upsampler_weights(w1, w2)
    b1 = torch.randn(64, device=device, generator=g) * 0.05
    lw, lb = torch.ones(64, device=device), torch.zeros(64, device=device)
    b2 = torch.randn(32, device=device, generator=g) * 0.05
    out = {"bound": "hbm", "kernel": "upsample_fused_kernel", "peak": 8000.0, "unit": "GB/s", "batch": 8, "dtype": "bf16"}
    for grid, key in ((16, "sam256"), (64, "sam1024")):
        B, n_buf = 8, (2 if grid == 16 else 12)
        srcs = [torch.randn(B, grid * grid, 256, device=device, generator=g).to(torch.bfloat16) for _ in range(n_buf)]
        ups = [torch.empty(B, 32, 4 * grid, 4 * grid, dtype=torch.bfloat16, device=device) for _ in range(n_buf)]

        def run(i):
            ops.lib().call("mp_mask_upsample_fused_bf16", srcs[i % n_buf].data_ptr(), w1p.data_ptr(), b1.data_ptr(), lw.data_ptr(),
                           lb.data_ptr(), w2p.data_ptr(), b2.data_ptr(), None, ups[i % n_buf].data_ptr(), None, B, grid, grid, 1e-6,
                           torch.cuda.current_stream().cuda_stream)
        for i in range(20):
            run(i)
        torch.cuda.synchronize()
        # 24 launches (rotating buffers) captured into one HIP graph and replayed: the time per launch is the kernel's, not the host's
        # launch cadence (plain back-to-back launches from Python read 3 us longer per launch); fallback: the plain loop
        per, graph = 24, None
        try:
            cs = torch.cuda.Stream(device=device)
            cs.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(cs):
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph, stream=cs):
                    for i in range(per):
                        run(i)
            torch.cuda.current_stream().wait_stream(cs)
            torch.cuda.synchronize()
        except Exception:
            graph = None
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 40
        # un-timed replays first (~50 ms of this same kernel): the first ~15 ms after an idle gap run at a lower clock — scripts/lab/ups_lab.hip
        # reads 20.7 us per launch for its first 630 launches and 17.3 us for the same launches later in the process — and a 19 ms
        # measurement taken right after the allocations above would sit inside that ramp (the copy floor below gets the same treatment)
        for _ in range(120):
            if graph is not None:
                graph.replay()
            else:
                for i in range(per):
                    run(i)
        e0.record()
        for _ in range(reps):
            if graph is not None:
                graph.replay()
            else:
                for i in range(per):
                    run(i)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / (reps * per)
        tokens = B * grid * grid
        nbytes = tokens * 256 * 2 + (256 * 256 + 128 * 64) * 2 + tokens * 16 * 32 * 2
        out[key] = {"geometry": f"{grid}x{grid} tokens ({grid * 16}-px SAM)", "us_per_launch": round(us, 2), "algorithmic_MB": round(nbytes / 1e6, 2),
                    "achieved": round(nbytes / us / 1e3, 1), "frac": round(nbytes / (us * 1e-6) / 8.0e12, 4)}
    out["achieved"], out["frac"] = out["sam1024"]["achieved"], out["sam1024"]["frac"]
    # what a TRIVIAL kernel gets for the same byte volume on this GPU, measured the same way: a plain bf16 copy of 25.2 MB (25.2 read + 25.2
    # written = the 50.3 MB of the sam1024 launch), rotating buffers, graph replay — the practical ceiling for a 50 MB launch, which the
    # 8 TB/s figure is not (launch ramp + drain are a fifth of a ~10 us kernel)
    try:
        nb = 12
        ca = [torch.randn(8 * 4096 * 256 * 3 // 2, device=device, generator=g).to(torch.bfloat16) for _ in range(nb)]
        cb = [torch.empty_like(x) for x in ca]
        for i in range(5):
            cb[i % nb].copy_(ca[i % nb])
        torch.cuda.synchronize()
        cs = torch.cuda.Stream(device=device)
        cs.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(cs):
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, stream=cs):
                for i in range(24):
                    cb[i % nb].copy_(ca[i % nb])
        torch.cuda.current_stream().wait_stream(cs)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for _ in range(120):
            graph.replay()
        e0.record()
        for _ in range(40):
            graph.replay()
        e1.record()
        torch.cuda.synchronize()
        cus = e0.elapsed_time(e1) * 1e3 / (40 * 24)
        cbytes = 2 * ca[0].numel() * 2
        out["copy_floor"] = {"what": "torch bf16 copy, 25.2 MB -> 25.2 MB (the sam1024 launch's 50.3 MB), same timing method",
                             "us_per_launch": round(cus, 2), "achieved": round(cbytes / cus / 1e3, 1), "frac_of_8TBps": round(cbytes / (cus * 1e-6) / 8.0e12, 4),
                             "upsampler_vs_copy": round(cus / out["sam1024"]["us_per_launch"], 4)}
    except Exception as e:
        out["copy_floor"] = {"error": f"{type(e).__name__}: {e}"}
    out["note"] = ("24 launches per captured HIP graph, 120 un-timed then 40 timed replays bracketed by HIP events (rotating inputs beyond the Infinity "
                   "Cache at the large geometry); the model runs the sam256 geometry, which is latency-bound at 3.29 MB; round 4: both row parities of a "
                   "token group in one workgroup, tokens requested behind the weight DMA, GEMM2 transposed -> whole-line non-temporal stores (DESIGN.md "
                   "section 3.2; timeline / ablation / PMC evidence: profiles/r04_upsampler_*)")
    return out


def lora_secondary(args, device, ds_config, synthetic_batch, rank, steps=8, warmup=3, secondary=None):
    """Secondary object of the default line: scripts/train_stage3.sh's configuration (the one every shipped script trains: dense Llama-7B,
    LoRA r = 8 / alpha 16 / dropout 0.05 on gate / up / down_proj, mask decoder + text_hidden_fcs trainable) = the whole decoder backward in
    the step, measured in the same process after the headline leg (8 steps after 3) so the driver's run carries it."""
    from medplib_amd import engine
    from medplib_amd.model.config import MedPLIBConfig
    from medplib_amd.model.medplib import LISAForCausalLM
    torch.manual_seed(1234)
    cfg = MedPLIBConfig.medplib_7b(num_hidden_layers=args.layers, moe_enable=False)
    model = LISAForCausalLM(cfg, device=device).train()
    extra = {}
    if secondary is not None:
        # the dense model before its adapters exist: BASELINE configs[1] (VQA-only forward, MoE disabled, batch 4) and evaluate()'s decode
        try:
            secondary["configs"]["1"] = dict(forward_rate(model, vqa_batch(cfg, 4, device, seed=42), 4, 63 + cfg.clip_num_patches, 4),
                                            what="BASELINE configs[1]: MedPLIB-7B bf16 VQA-only forward (MoE disabled), batch 4, CE loss")
            secondary["decode"]["dense"] = decode_rate(model, device)
        except Exception as e:
            secondary["configs"]["1"] = {"error": f"{type(e).__name__}: {e}"}
    lora = model.enable_lora(lora_r=8, lora_alpha=16, lora_dropout=float(os.environ.get("MP_BENCH_LORA_DROPOUT", "0.05")), lora_target_modules="gate_proj,up_proj,down_proj",
                             sft_modules="mask_decoder,text_hidden_fcs")
    for n, p in zip(lora.names, lora.params):                # B = 0 at initialisation would make half the gradients trivially zero
        if "lora_B" in n:
            p.data.normal_(0, 0.01)
    model.towers_run_ahead = not args.towers_in_order
    eng, _, _, _ = engine.initialize(model=model, model_parameters=model.trainable_parameters(), config=ds_config)
    batch = synthetic_batch(cfg, args.batch, device, seed=42 + rank)
    for _ in range(warmup):
        out = eng(**batch); eng.backward(out); eng.step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        out = eng(**batch); eng.backward(out); eng.step()
    dt_issue = time.perf_counter() - t0
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    res = {"workload": "MedPLIB-7B dense stage-III training step WITH LoRA (r=8 on gate/up/down_proj, dropout 0.05: scripts/train_stage3.sh), "
                       f"whole decoder backward, per-GPU batch {args.batch}", "steps": steps, "warmup": warmup,
           "ms_per_step": round(dt / steps * 1e3, 2), "host_issue_ms_per_step": round(dt_issue / steps * 1e3, 2), "samples_per_s": round(args.batch * steps / dt, 2),
           "model_tflops_per_gpu": round(((FWD_TFLOP_PER_SAMPLE + 8.66) * args.batch - pruned_tflop_per_step(model, cfg, True)[0]) * steps / dt, 1),
           "last_layer_mlp_rows": pruned_tflop_per_step(model, cfg, True)[1] or "all",
           "trainable_params": eng.optimizer.numel, "loss_last": float(out["loss"].detach())}
    model.sync_side_streams(); torch.cuda.synchronize()
    del eng, model, lora, out
    torch.cuda.empty_cache()
    return res


def claim_stdout():
    """stdout carries ONE JSON line and nothing else: fd 1 is pointed at stderr for the life of the process (RCCL prints a version banner
    to the C stdout of every rank, flushed at exit — i.e. after the line) and the returned function writes to the original stdout."""
    sys.stdout.flush()
    real = os.dup(1)
    os.dup2(2, 1)

    def emit(text):
        os.write(real, (text + "\n").encode())
    return emit


def ensure_ranks(want, dry):
    """`--gpus N` must mean N ranks.  No launcher in the environment and N > 1: replace this process by
    `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port <free> bench.py <same flags>`
    (the reference's launch line is `deepspeed --include localhost:0..7 train_ds_medplib.py`, scripts/train_stage3.sh).  A launcher
    that started a different number of ranks is an error, not a smaller benchmark."""
    world = os.environ.get("WORLD_SIZE")
    if world is None and want > 1:
        if not dry and torch.cuda.device_count() < want:
            sys.exit(f"bench.py: --gpus {want} but only {torch.cuda.device_count()} GPU(s) visible")
        import socket
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={want}", "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        sys.stdout.flush(); sys.stderr.flush()
        os.execvpe(cmd[0], cmd, env)
    if int(world or 1) != want:
        sys.exit(f"bench.py: --gpus {want} but the launcher started WORLD_SIZE={world} rank(s); refusing to report n_gpus != requested")


def dry_main(args, emit):
    """The multi-rank protocol of this file without a GPU: gloo ranks, a stand-in flat gradient bucket through the engine's
    launch_grad_reduce / wait_grad_reduce, the barrier + MAX-over-ranks timing, one JSON line from rank 0."""
    from medplib_amd import engine
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    if world > 1:
        dist.init_process_group(backend="gloo")
    tail = torch.nn.Linear(64, 32)
    eng, _, _, _ = engine.initialize(model=tail, model_parameters=list(tail.parameters()), config={"optimizer": {"params": {"lr": 1e-3}}})
    ok = True
    epx, ep_ok, E, d_, cap_ = None, True, 2, 16, 24
    if args.ep:
        # `--ep`'s twin on CPU ranks: DeepSpeed's group shapes, the host-side capacity agreement, and per step one dispatch / combine round
        # trip of seeded rows through ExpertParallel (padded slabs or, with --ep-variable, routed rows only) that every rank checks
        from medplib_amd import expert_parallel as EPM
        if world % args.ep:
            sys.exit(f"bench.py: --ep {args.ep} does not divide {world} ranks")
        E = max(2, args.ep)
        if world > 1:
            group, _ = EPM.build_groups(args.ep)
            host_group = EPM.build_host_group(args.ep)
        else:
            group = host_group = None
        epx = EPM.ExpertParallel(group, args.ep if world > 1 else 1, E, host_group=host_group, variable_split=args.ep_variable)

    def ep_round_trip(k):
        """rank r routes (r + k) % cap_ + 1 rows to every expert; expert e returns (e + 2) * row; every row must come home"""
        g = torch.Generator().manual_seed(1000 * rank + k)
        capx = epx.exchange_capacity(cap_, key=k)
        kept = torch.tensor([(rank + k + e) % cap_ + 1 for e in range(E)], dtype=torch.int32)
        buf = torch.zeros(E, capx + 1, d_)
        for e in range(E):
            buf[e, :int(kept[e])] = torch.randn(int(kept[e]), d_, generator=g)
        sent = buf.clone()
        recv, counts = epx.dispatch(buf, kept)
        y = torch.zeros(epx.ep, epx.E_local, capx, d_)
        for s_ in range(epx.ep):
            for el, ge in enumerate(epx.local_expert_ids()):
                n = int(counts[s_, el])
                y[s_, el, :n] = recv[s_, el, :n] * (ge + 2.0)
        out = epx.combine(y)
        return all(torch.allclose(out[e, :int(kept[e])], sent[e, :int(kept[e])] * (e + 2.0)) for e in range(E))

    def step(k):
        nonlocal ok, ep_ok
        eng.optimizer.flat_grad.fill_(float(rank + 1 + k))
        eng.launch_grad_reduce(); eng.wait_grad_reduce()
        ok &= bool(torch.all(eng.optimizer.flat_grad == sum(r + 1 + k for r in range(world))))
        if epx is not None:
            ep_ok &= ep_round_trip(k)
    for k in range(args.warmup):
        step(k)
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for k in range(args.steps):
        step(k)
    if world > 1:
        dist.barrier()
    tmax = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    allok = torch.tensor([int(ok and ep_ok)], dtype=torch.int64)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(allok, op=dist.ReduceOp.MIN)
    if rank == 0:
        dt = tmax.item()
        extra = {}
        if epx is not None:
            extra["ep"] = {"ep_size": epx.ep, "replicas": world // max(args.ep, 1), "experts": E, "variable_split": bool(args.ep_variable),
                           "exchanges_per_step": epx.stats["exchanges"] / max(args.steps + args.warmup, 1),
                           "a2a_bytes_sent_per_exchange": epx.stats["bytes_sent"] / max(epx.stats["exchanges"], 1),
                           "round_trips_correct_on_every_rank": bool(allok.item())}
        emit(json.dumps({**extra, "metric": "dry run: gradient-bucket all-reduce steps/s on CPU ranks (gloo)", "dry": True, "value": round(args.steps / dt, 3),
                          "unit": "steps/s", "n_gpus": world, "rccl_ranks": None, "ranks": dist.get_world_size() if world > 1 else 1,
                          "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
                          "higher_is_better": True, "scaling": "weak", "bucket_sums_correct": ok,
                          "config": {"workload": f"{eng.optimizer.numel}-element fp32 bucket, backend gloo",
                                     "parallelism": (f"ep{args.ep} x dp{world // args.ep}" if args.ep else f"dp{world}")}}))
    if world > 1:
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=8, help="per-GPU micro-batch (BASELINE config: 8)")
    ap.add_argument("--layers", type=int, default=32, help="debug only; the reported config is 32")
    ap.add_argument("--experts", type=int, default=2, help="debug only: experts per MoE layer (the reported config: 2)")
    ap.add_argument("--top-k", type=int, default=1, choices=(1, 2), help="debug only: DeepSpeed top-k gating (the reported config: top-1)")
    ap.add_argument("--use-residual", action="store_true", help="debug only: DeepSpeed MoE(use_residual=True)")
    ap.add_argument("--lora", action="store_true",
                    help="secondary line: scripts/train_stage3.sh's configuration (dense Llama-7B, LoRA r = 8 / alpha 16 / dropout 0.05 on "
                         "gate/up/down_proj, mask decoder + text_hidden_fcs trainable) = the whole decoder backward in the step; the "
                         "contract's default line is BASELINE configs[3] (LoRA off)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-lora-line", action="store_true", help="skip the secondary LoRA measurement (`lora_stage3`) of the default line")
    ap.add_argument("--no-secondary", action="store_true", help="skip the `configs` (BASELINE configs[1], [2], [4] forwards) and `decode` objects of the default line")
    ap.add_argument("--no-kernel-timer", action="store_true")
    ap.add_argument("--no-live-traffic", action="store_true", help="skip the two rocprofv3 child passes that measure `roofline.traffic` in this run "
                    "(the committed profile's figure is reported instead, marked as such)")
    ap.add_argument("--fold-input-norm", action="store_true",
                    help="config.fold_input_norm: both RMSNorms of every frozen MoE decoder layer folded into their consumer GEMMs (A/B; moves a rounding point)")
    ap.add_argument("--towers-in-order", action="store_true",
                    help="debug / A-B: the frozen CLIP tower of a step queues behind the previous step's decoder instead of starting on its own "
                         "stream when the step is issued (model.towers_run_ahead, the default since round 3: +2.6 %% samples/s)")
    ap.add_argument("--roofline-steps", type=int, default=6,
                    help="extra un-timed steps AFTER the timed region in which the dominant GEMM is measured unshared (towers in order, SAM "
                         "encoder and mask tail on the decoder's stream): `roofline`; the timed region's own figure is `roofline_timed_region`")
    ap.add_argument("--no-side-streams", action="store_true",
                    help="debug only: SAM encoder and mask tail on the decoder's stream (what the side streams buy; what they cost the GEMM launches they run beside)")
    ap.add_argument("--host-inputs", action="store_true",
                    help="images and masks start every step in pageable host memory (the reference's dict_to_cuda per batch): the "
                         "PCIe-inclusive rate quoted in DESIGN.md; `value` of the contract is the default, HBM-resident run")
    ap.add_argument("--ep", type=int, default=0,
                    help="expert-parallel size: run BASELINE configs[4] instead (MedPLIB-ICL separate mode, 3 in-context (image, mask) pairs + query, "
                         "mm_token_compress 576->256, E=2 top-1 experts sharded over ep ranks with the all-to-all exchange, per-GPU batch 4) as "
                         "ep x (gpus / ep) replicas: scripts/train_medplib_icl.sh:15-43, medplib_moe_llama.py:604-614 (ep_size).  8 GPUs: --ep 2")
    ap.add_argument("--ep-variable", action="store_true", help="with --ep: routed rows only instead of capacity-padded slabs (a host read per layer)")
    ap.add_argument("--ep-comm", default="torch", choices=("torch", "capi"),
                    help="with --ep: the exchange through torch.distributed's all_to_all_single (default) or the C ABI's mp_alltoall_tokens")
    ap.add_argument("--dry", action="store_true", help="CPU ranks over gloo, stand-in gradient bucket: the launch / timing / one-line protocol only")
    args = ap.parse_args()
    if args.ep and args.gpus % args.ep:
        sys.exit(f"bench.py: --ep {args.ep} does not divide --gpus {args.gpus}")
    if args.ep and "--batch" not in " ".join(sys.argv):
        args.batch = 4                                        # configs[4]: batch 4 per GPU
    ensure_ranks(args.gpus, args.dry)
    emit = claim_stdout()
    try:
        return dry_main(args, emit) if args.dry else gpu_main(args, emit)
    except BaseException as e:
        # a rank that dies says which one it was and why, on its own stderr, before the launcher tears the others down
        if not isinstance(e, SystemExit) or e.code not in (0, None):
            import traceback
            sys.stderr.write(f"[bench rank {os.environ.get('RANK', '0')}/{os.environ.get('WORLD_SIZE', '1')} "
                             f"local {os.environ.get('LOCAL_RANK', '0')}] failed: {type(e).__name__}: {e}\n{traceback.format_exc()}")
            sys.stderr.flush()
        raise


def gpu_main(args, emit):

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    parity_failed = False
    # debug, one-GPU boxes: MP_BENCH_SHARE_GPU=1 puts every rank on cuda:0 and MP_BENCH_BACKEND=gloo carries the collectives (RCCL refuses two
    # ranks on one device) — N real processes through the multi-rank code path with real kernels (tests/test_gpu_bench_dp.py)
    backend = os.environ.get("MP_BENCH_BACKEND", "nccl")
    if os.environ.get("MP_BENCH_SHARE_GPU") == "1":
        local = 0
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    # the GPU leg's host work is index planning on 8 x 64 ids: a handful of threads per rank (N ranks x every core would
    # oversubscribe the host with spinning OpenMP pools); the cpu_baseline leg sets its own thread count
    torch.set_num_threads(max(1, min(8, (os.cpu_count() or 8) // max(world, 1))))
    # MP_BENCH_FORCE_DIST=1 (debug, one-GPU boxes): take the multi-rank code path on a ONE-rank RCCL group — process group, the C-ABI
    # communicator, the gradient bucket on the communication stream with its event timing, barriers, the MAX over ranks
    force_dist = os.environ.get("MP_BENCH_FORCE_DIST") == "1"
    if world > 1 or force_dist:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if world == 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29631")
            os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
        # an explicit collective timeout (a peer that died or never joined ends the run with an error naming the collective, not a hang
        # until the driver's own limit): MP_BENCH_NCCL_TIMEOUT seconds, default 600
        import datetime
        os.environ.setdefault("TORCH_NCCL_ASYNC_ERROR_HANDLING", "1")
        dist.init_process_group(backend=backend, timeout=datetime.timedelta(seconds=int(os.environ.get("MP_BENCH_NCCL_TIMEOUT", "600"))))
    rccl_ranks = None
    if (world > 1 or force_dist) and backend == "nccl":
        # what RCCL itself connected: a communicator through the C ABI (bootstrapped with a unique id that travels over the process group),
        # its own rank count (ncclCommCount) and a SUM of ones over it
        try:
            from medplib_amd.comm import RcclComm
            cc = RcclComm()
            one = torch.ones(1, dtype=torch.float32, device=device)
            cc.all_reduce_(one)
            torch.cuda.synchronize()
            rccl_ranks = {"ncclCommCount": cc.count()[0], "sum_of_ones": float(one.item())}
            cc.close()
        except Exception as e:            # the line says so instead of guessing
            rccl_ranks = {"error": f"{type(e).__name__}: {e}"}

    from medplib_amd import engine, ops
    from medplib_amd.model.config import MedPLIBConfig
    from medplib_amd.model.medplib import MedPLIBForCausalLM

    torch.manual_seed(1234)          # the randomly initialised trainable tail is the same on every rank and in every run
    epx = None
    if args.ep:
        # BASELINE configs[4]: the ICL model (token compressor + mask encoder + E = 2 experts), experts sharded over `ep` consecutive ranks
        from medplib_amd import expert_parallel as EPM
        cfg = icl_config(args.layers)
        model = MedPLIBForCausalLM(cfg, device=device).train()
        if world > 1 or force_dist:
            ep_size = args.ep if world > 1 else 1
            group, _ = EPM.build_groups(ep_size)
            host_group = EPM.build_host_group(ep_size)
            capi = None
            if args.ep_comm == "capi":
                from medplib_amd.comm import RcclComm
                capi = RcclComm(group=group)
            epx = EPM.ExpertParallel(group, ep_size, cfg.num_experts, host_group=host_group, capi_comm=capi, variable_split=args.ep_variable)
        else:
            epx = EPM.ExpertParallel(None, 1, cfg.num_experts, variable_split=args.ep_variable)     # one GPU, no process group: the ep code path on a trivial group
        model.model.llm.enable_expert_parallel(epx)
        args.no_cpu_baseline = args.no_lora_line = True
    elif args.lora:
        from medplib_amd.model.medplib import LISAForCausalLM
        cfg = MedPLIBConfig.medplib_7b(num_hidden_layers=args.layers, moe_enable=False)
        model = LISAForCausalLM(cfg, device=device).train()
        lora = model.enable_lora(lora_r=8, lora_alpha=16, lora_dropout=float(os.environ.get("MP_BENCH_LORA_DROPOUT", "0.05")), lora_target_modules="gate_proj,up_proj,down_proj",
                                 sft_modules="mask_decoder,text_hidden_fcs")
        for n, p in zip(lora.names, lora.params):            # B = 0 at initialisation would make half the gradients trivially zero
            if "lora_B" in n:
                p.data.normal_(0, 0.01)
        args.no_cpu_baseline = True                          # the host leg and the parity object belong to the default configuration
    else:
        cfg = MedPLIBConfig.medplib_7b(num_hidden_layers=args.layers, num_experts=args.experts, top_k_experts=args.top_k,
                                       use_residual=args.use_residual, fold_input_norm=bool(args.fold_input_norm))
        if (args.experts, args.top_k, args.use_residual) != (2, 1, False):
            args.no_cpu_baseline = True                      # the host leg and the parity object belong to the reported configuration
        model = MedPLIBForCausalLM(cfg, device=device).train()
    ds_config = {"train_micro_batch_size_per_gpu": args.batch, "gradient_accumulation_steps": 1,
                 "optimizer": {"type": "AdamW", "params": {"lr": 3e-4, "weight_decay": 0.0, "betas": (0.9, 0.95)}},
                 "gradient_clipping": 1.0,
                 "scheduler": {"type": "WarmupDecayLR", "params": {"total_num_steps": 10000, "warmup_min_lr": 0,
                                                                     "warmup_max_lr": 3e-4, "warmup_num_steps": 100,
                                                                     "warmup_type": "linear"}}}
    # the synthetic images are resident before the loop starts, so the frozen towers of a step start as soon as the step is issued (beside the
    # previous step's last decoder layers) instead of queueing behind them.  The decoder GEMM launches that share the chip with the towers
    # stretch, so the kernel's own roofline figure is taken from an unshared sample after the timed region (below).
    model.towers_run_ahead = not args.towers_in_order and not args.host_inputs and not args.no_side_streams
    if args.no_side_streams:
        model.sam_side_stream = False
        ds_config["overlap_mask_tail"] = 0
    if force_dist and os.environ.get("MP_BENCH_NO_REDUCE") != "1":
        ds_config["reduce_single_rank"] = 1
    if os.environ.get("MP_BENCH_COMM"):                  # "rccl_capi": the gradient bucket through the library's own mp_allreduce_bucket
        ds_config["comm_backend"] = os.environ["MP_BENCH_COMM"]
    eng, _, _, _ = engine.initialize(model=model, model_parameters=model.trainable_parameters(), config=ds_config)
    seq_len = 639
    if args.ep:
        batch, seq_len = synthetic_icl_batch(cfg, args.batch, device, seed=42 + rank)
    else:
        batch = synthetic_batch(cfg, args.batch, device, seed=42 + rank)

    host_batch = None
    if args.host_inputs:
        host_batch = {k: (v.cpu() if torch.is_tensor(v) else [m.cpu() for m in v] if k == "masks_list" else v) for k, v in batch.items()}

    def step():
        if host_batch is not None:       # a fresh host -> device copy of every image / mask tensor, inside the timed region
            b = {k: (v.to(device) if torch.is_tensor(v) else [m.to(device) for m in v] if k == "masks_list" else v)
                 for k, v in host_batch.items()}
            out = eng(**b)
        else:
            out = eng(**batch)
        eng.backward(out)                 # the output dict: its loss tensors live on the mask tail's stream, nothing here reads them
        eng.step()
        return out

    # debug: run the step's main stream at high priority, so the CLIP tower / decoder get the CUs before the side streams' kernels
    hp_ctx = torch.cuda.stream(torch.cuda.Stream(device=device, priority=-1)) if os.environ.get("MP_BENCH_HIPRIO") == "1" else None
    if hp_ctx is not None:
        torch.cuda.synchronize()
        hp_ctx.__enter__()
    for _ in range(args.warmup):
        out = step()
    torch.cuda.synchronize()
    loss0 = float(out["loss"].detach()) if args.warmup else float("nan")
    if rank == 0:
        print(f"[bench] warm-up done, loss {loss0:.4f}", file=sys.stderr, flush=True)

    timer = None
    if not args.no_kernel_timer:
        timer = ops.KernelTimer(sample_every=23)         # every 23rd GEMM launch (coprime to the layer's GEMM period): ~290 samples over 20 steps
        ops.GEMM_TIMER = timer
    if epx is not None:
        epx.stats = {"exchanges": 0, "bytes_sent": 0}
        epx.timing, epx.sample_every = [], 5             # every 5th exchange bracketed by HIP events on its stream
    eng.enable_bucket_timing()           # HIP-event pairs around the tail backward and every gradient bucket (a few events per step)
    if world > 1 or force_dist:
        dist.barrier()
    torch.cuda.synchronize()
    mark = (lambda tag: ops.lib().call("mp_profile_marker", tag, torch.cuda.current_stream().cuda_stream)) if os.environ.get("MP_BENCH_MARKERS") == "1" else (lambda tag: None)
    mark(1)                              # (profiling runs only, scripts/r04_profiles.sh: cut marks for scripts/rocpd_stats.py; the device is idle here)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    dt_issue = time.perf_counter() - t0     # the host's time to ISSUE the K steps (no device wait inside): equal to `dt` = the step is host-bound
    if world > 1 or force_dist:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    mark(2)
    ops.GEMM_TIMER = None
    ep_obj = None
    if epx is not None:
        us = [s0.elapsed_time(e0) * 1e3 for _, s0, e0 in epx.timing]
        n_moe = len(cfg.moe_layer_set())
        ep_obj = {"ep_size": epx.ep, "replicas": max(world // max(epx.ep, 1), 1), "experts": cfg.num_experts, "experts_per_rank": epx.E_local,
                  "transport": "mp_alltoall_tokens (C ABI, grouped ncclSend/ncclRecv)" if epx.capi_comm is not None else "torch.distributed all_to_all_single (RCCL)",
                  "variable_split": bool(epx.variable_split), "moe_layers": n_moe,
                  "exchanges_per_step": round(epx.stats["exchanges"] / args.steps, 2),
                  "a2a_bytes_per_layer": round(epx.stats["bytes_sent"] / max(args.steps * n_moe, 1)),       # sent by this rank, dispatch + combine
                  "a2a_bytes_per_exchange": round(epx.stats["bytes_sent"] / max(epx.stats["exchanges"], 1)),
                  "a2a_us": round(sum(us) / len(us), 1) if us else None, "a2a_us_max": round(max(us), 1) if us else None,
                  "sampled_exchanges": len(us)}
        epx.timing = None
    dp_bucket = eng.bucket_timing_summary(args.steps)        # the timed steps only: the roofline / LoRA steps below must not count
    # how the trainable fp32 tail ran in the timed steps: as the two program launches (csrc/tail_program.hip) or op by op (MP_TAIL_PROGRAM=0)
    tail_obj = None
    try:
        dec_ = model.model.visual_model.mask_decoder
        run_ = getattr(dec_, "_runner", None)
        if dec_.use_program and run_ is not None and run_._cache:
            pg = next(iter(run_._cache.values()))
            tail_obj = {"form": "program: one launch forward, one backward" + (" (fused upsampler's gradients finished in-program)" if run_.fused_upsampler else ""),
                        "grid": run_.grid, "forward": {"ops": len(pg.fwd_packed[0]), "phases": len(pg.fwd_packed[2])},
                        "backward": {"ops": len(pg.bwd_packed[0]), "phases": len(pg.bwd_packed[2])},
                        "workspace_MB": round((pg.fwd_bytes + pg.bwd_bytes) / 1e6, 1), "barrier_gave_up": not pg.check_sync()}
        else:
            tail_obj = {"form": "op by op (autograd Functions over mp_sgemm_f32 and the fp32 row kernels)"}
    except Exception as e:
        tail_obj = {"error": f"{type(e).__name__}: {e}"}
    eng.disable_bucket_timing()
    tmax = torch.tensor([dt], dtype=torch.float64, device=device)
    if world > 1 or force_dist:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = tmax.item()

    # ---- the dominant kernel measured UNSHARED: a few more steps (not part of `value`) with the towers in order and the SAM encoder / mask
    # tail on the decoder's own stream, so nothing else is resident on the CUs while a sampled GEMM launch runs
    timer_u = None
    if timer is not None and args.roofline_steps > 0 and not args.lora:
        model.sync_side_streams(); torch.cuda.synchronize()
        keep = (model.towers_run_ahead, model.sam_side_stream, model.tail_side_stream)
        model.towers_run_ahead, model.sam_side_stream, model.tail_side_stream = False, False, False
        step(); torch.cuda.synchronize()                       # one step for the stream change to settle
        timer_u = ops.KernelTimer(sample_every=7)
        ops.GEMM_TIMER = timer_u
        for _ in range(args.roofline_steps):
            step()
        torch.cuda.synchronize()
        ops.GEMM_TIMER = None
        model.towers_run_ahead, model.sam_side_stream, model.tail_side_stream = keep

    def roofline_of(timer, n_steps):
        """The `roofline` object from one KernelTimer: the dominant bf16 GEMM tile kernel (most GPU time among the sampled launches),
        the other tile kernel and all bf16 GEMM launches beside it."""
        roof = None
        if timer is not None:
            # every 23rd GEMM launch of the timed steps is bracketed by HIP events on the launch stream (ops.KernelTimer)
            # the dominant kernel alone (mp_gemm_last_kernel tells which kernel a launch went to), then all bf16 GEMM launches
            fams = {256: "gemm256v3_bf16_nt_kernel", 320: "gemm320_bf16_nt_kernel"}
            summ = {k: timer.summary(k) for k in fams}
            a_flops, a_ms, a_sampled, a_launches, a_all = timer.summary()

            def est_ms(k):            # time per run estimated from the sampled launches: all launches x the sampled average
                f, m, n, l, _ = summ[k]
                return l * m / n if n else 0.0
            dom = max(fams, key=est_ms)                 # the dominant kernel = the family with the most GPU time in the timed steps
            flops, ms, sampled, launches, all_flops = summ[dom]
            if sampled == 0:          # very short debug runs: no sampled launch went to either tile kernel
                flops, ms, sampled, launches, all_flops = a_flops, a_ms, a_sampled, a_launches, a_all
            if os.environ.get("MP_BENCH_SHAPES"):       # debug: the sampled launches grouped by their algorithmic work (= by shape)
                by = {}
                for w, s0, e0, kern in timer.records:
                    by.setdefault((kern, round(w / 1e9, 1)), []).append(s0.elapsed_time(e0) * 1e3)
                for (kern, gf), v in sorted(by.items()):
                    print(f"[bench] gemm{kern} {gf:8.1f} GFLOP: {len(v):4d} samples, avg {sum(v) / len(v):7.1f} us, "
                          f"min {min(v):7.1f}, {gf / (sum(v) / len(v)) * 1e3 / 1e3:7.1f} TF/s", file=sys.stderr)
            achieved = flops / (max(ms, 1e-9) * 1e-3) / 1e12
            a_ach = a_flops / (max(a_ms, 1e-9) * 1e-3) / 1e12
            roof = {"bound": "mfma", "kernel": fams[dom],
                    "achieved": round(achieved, 1), "peak": MFMA_BF16_PEAK_TFLOPS,
                    "unit": "TFLOP/s", "frac": round(achieved / MFMA_BF16_PEAK_TFLOPS, 4), "traffic": None,
                    "launches_per_step": launches // n_steps, "sampled_launches": sampled,
                    "avg_launch_us": round(ms * 1e3 / max(sampled, 1), 2),
                    "gemm_tflop_per_step": round(all_flops / n_steps / 1e12, 2),
                    "gemm_ms_per_step": round(all_flops / n_steps / (achieved * 1e12) * 1e3, 2),
                    "all_bf16_gemms": {"kernels": "gemm256v3_bf16_nt_kernel + gemm320_bf16_nt_kernel + gemm_bf16_nt_kernel", "achieved": round(a_ach, 1),
                                       "frac": round(a_ach / MFMA_BF16_PEAK_TFLOPS, 4), "launches_per_step": a_launches // n_steps,
                                       "sampled_launches": a_sampled, "avg_launch_us": round(a_ms * 1e3 / max(a_sampled, 1), 2),
                                       "tflop_per_step": round(a_all / n_steps / 1e12, 2)}}
            # the expert GEMMs are credited with the rows they PROCESSED (the device-side `kept` counts, read back after the region): a
            # capacity-dropped token is work the kernel skips, not work it did
            if timer.kept_rows:
                per_layer = [round(sum(v) / len(v)) for _, v in sorted((t, v) for t, v in timer.kept_rows.items() if t is not None)]
                roof["expert_rows"] = {"credit": "kept rows per launch (device-side counts), not tokens", "tokens": args.batch * seq_len * cfg.top_k_experts,
                                       "kept_rows_per_layer": per_layer, "kept_rows_min": min(per_layer) if per_layer else None}
            tw = [timer.summary(1000 + k) for k in (320, 256, 128)]        # the frozen towers' launches (throughput tiles), kept apart
            tw_f, tw_ms, tw_n, tw_l = sum(t[0] for t in tw), sum(t[1] for t in tw), sum(t[2] for t in tw), sum(t[3] for t in tw)
            if tw_n:
                roof["tower_gemms"] = {"note": "CLIP tower / projector / SAM encoder GEMMs (320-row tiles whenever eligible: few, fat workgroups, "
                                               "priced by CU x time beside the decoder, not by latency)", "launches_per_step": tw_l // n_steps,
                                       "sampled_launches": tw_n, "avg_launch_us": round(tw_ms * 1e3 / tw_n, 2),
                                       "achieved": round(tw_f / (max(tw_ms, 1e-9) * 1e-3) / 1e12, 1)}
            for k in fams:             # the other tile kernel beside the dominant one (320-row tiles: the dense projections and the experts'
                if k == dom:           # gate|up; 256x256 tiles: the experts' down projection with the combine epilogue, CLIP's qkv / fc2)
                    continue
                t_flops, t_ms, t_sampled, t_launches, t_all = summ[k]
                if t_sampled:
                    t_ach = t_flops / (max(t_ms, 1e-9) * 1e-3) / 1e12
                    roof[fams[k]] = {"achieved": round(t_ach, 1), "frac": round(t_ach / MFMA_BF16_PEAK_TFLOPS, 4),
                                     "launches_per_step": t_launches // n_steps, "sampled_launches": t_sampled,
                                     "avg_launch_us": round(t_ms * 1e3 / t_sampled, 2), "tflop_per_step": round(t_all / n_steps / 1e12, 2)}
            # HBM-side bytes per launch of the dominant kernel come from PMC passes (FETCH_SIZE / WRITE_SIZE in separate
            # rocprofv3 runs of this same command, scripts/bench_pmc.sh), which cannot be taken from inside the process: the
            # committed summary is reported with its provenance.  (FETCH_SIZE counts L2 misses incl. Infinity-Cache hits.)
            pdir = os.path.join(ROOT, "profiles")
            import glob
            cands = sorted(glob.glob(os.path.join(pdir, "r*_hbm_traffic.json")), key=lambda q: os.path.basename(q), reverse=True)
            tname = os.path.basename(cands[0]) if cands else None
            if tname:
                tjs = json.load(open(os.path.join(pdir, tname)))
                # (since the towers' GEMMs share the 320-row kernel's plain family, the decoder's launches are the epilogue families 1-3)
                tj = (tjs.get("gemm320_decoder") if dom == 320 else None) or tjs.get({256: "gemm256v3", 320: "gemm320"}[dom])
                if live is not None and "error" not in live and dom == 320 and not args.lora and not args.ep:
                    roof["traffic"] = live["read_bytes_per_launch"] + live["write_bytes_per_launch"]
                    roof["traffic_detail"] = {"kernel": fams[dom], "read_bytes_per_launch": live["read_bytes_per_launch"],
                                              "write_bytes_per_launch": live["write_bytes_per_launch"], "launches": live["launches"],
                                              # GRBM_GUI_ACTIVE / 8 XCDs / the launch's wall time in the same child pass: the shader clock the chip held
                                              # under the dominant kernel (the MFMA peak is quoted at 2.4 GHz)
                                              "shader_clocks_per_launch": live.get("shader_clocks_per_launch"), "effective_clock_ghz": live.get("effective_clock_ghz"),
                                              "source": "measured in THIS run: two child passes of this command (2 steps after 1) under rocprofv3 --kernel-trace --pmc "
                                                        "FETCH_SIZE | WRITE_SIZE, the decoder's 320-row launches (epilogue families 1-3), read = 2 x FETCH_SIZE per the "
                                                        f"gfx950 correction; {live['seconds']} s", "stale": False, "kernel_source_sha": kernel_source_sha()}
                    if live.get("effective_clock_ghz"):
                        # beside `frac` (against the peak quoted at the 2.4 GHz maximum clock): the same achieved rate against the MFMA peak AT THE
                        # CLOCK THE CHIP HELD under this kernel — what the schedule leaves on the table once the power-limited clock is taken out
                        roof["effective_clock_ghz"] = live["effective_clock_ghz"]
                        roof["frac_at_effective_clock"] = round(roof["achieved"] / (MFMA_BF16_PEAK_TFLOPS * live["effective_clock_ghz"] / 2.4), 4)
                elif tj and not args.lora and not args.ep:
                    now, then = kernel_source_sha(), tjs.get("kernel_source_sha")
                    roof["traffic"] = tj["read_bytes_per_launch"] + tj["write_bytes_per_launch"]
                    if live is not None and "error" in live:
                        roof["traffic_live_error"] = live["error"]
                    roof["traffic_detail"] = {"kernel": fams[dom], "read_bytes_per_launch": tj["read_bytes_per_launch"],
                                              "write_bytes_per_launch": tj["write_bytes_per_launch"],
                                              "source": f"profiles/{tname} (rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE over this same command in separate passes, "
                                                        "read = 2 x FETCH_SIZE per the gfx950 correction): a builder-side profile, NOT measured in this run "
                                                        "(PMC counters cannot be collected from inside the process)",
                                              # stale = the GEMM kernels' sources changed since the profile was taken (sha over KERNEL_SOURCES; profiles
                                              # older than round 4 carry no stamp and count as stale)
                                              "profile_kernel_source_sha": then, "kernel_source_sha": now, "stale": then != now}
        return roof

    live = None
    under_profiler = any(k.startswith("ROCPROF") for k in os.environ) or "rocprof" in os.environ.get("LD_PRELOAD", "")
    if (rank == 0 and world == 1 and timer is not None and not args.no_live_traffic and not args.no_cpu_baseline and not args.lora and not args.ep
            and not force_dist and not under_profiler):        # the full default line only (the A/B and profiling commands skip their host legs)
        # (the parent is idle here: its timed region and the unshared roofline steps are over; the children bring their own model)
        model.sync_side_streams(); torch.cuda.synchronize()
        live = live_traffic()
    if rank == 0:
        samples = world * args.batch * args.steps
        value = samples / dt
        roof_timed = roofline_of(timer, args.steps)
        roof = roofline_of(timer_u, args.roofline_steps) if timer_u is not None else roof_timed
        if timer_u is not None and roof is not None:
            roof["sample"] = (f"{args.roofline_steps} extra steps after the timed region with nothing else resident on the CUs (frozen towers in order, SAM "
                              "encoder and mask tail on the decoder's stream), every 7th GEMM launch bracketed by HIP events; the timed region, where "
                              "the next step's towers and the previous step's tail share the chip with the decoder, is `roofline_timed_region`")
        res = {
            "metric": "train samples/sec (img+64tok)", "value": round(value, 3), "unit": "samples/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 2), "higher_is_better": True,
            "host_issue_ms_per_step": round(dt_issue / args.steps * 1e3, 2),
            "scaling": "weak", "vs_baseline": None, "dtype": "bf16",
            "data": "synthetic" + (" (images / masks copied from pageable host memory every step)" if args.host_inputs else "")
                    + (" (frozen towers started ahead on their own streams)" if model.towers_run_ahead else ""),
            "config": {"workload": (("MedPLIB-7B dense stage-III training step WITH LoRA (r=8 on gate/up/down_proj, dropout 0.05: scripts/train_stage3.sh; "
                                    "whole decoder backward), " if args.lora else
                                    f"MedPLIB-7B-MoE stage-III training step (CE+BCE+Dice+Focal, LoRA off; E={args.experts} top-{args.top_k} experts x{args.layers} layers"
                                    f"{', use_residual' if args.use_residual else ''}), ") +
                                   "336x336 CLIP image + 256x256 SAM image + 64-token prompt (S=639 after splice), "
                                   f"per-GPU batch {args.batch}, DP={world}") if not args.ep else
                                  (f"BASELINE configs[4]: MedPLIB-ICL separate mode, 3 in-context (image, mask) pairs + query, mm_token_compress 576->256, mask encoder "
                                   f"64 tokens, E={cfg.num_experts} top-1 experts x{args.layers} layers sharded over ep={epx.ep} ranks with the expert all-to-all, stage-III-style "
                                   f"training step (CE+BCE+Dice+Focal, LoRA off), S={seq_len} after splice, per-GPU batch {args.batch}, {world} GPU(s)"),
                       "global_batch": world * args.batch, "seq_len": seq_len,
                       "parallelism": (f"ep{epx.ep} x dp{max(world // epx.ep, 1)}" if args.ep else f"dp{world}"),
                       "llm_layers": cfg.num_hidden_layers, "trainable_params": eng.optimizer.numel,
                       # decoder layers whose two RMSNorms ran folded into their consumer GEMMs in the timed steps (config.fold_input_norm; 0 = HF's rounding points)
                       "folded_norm_layers": int(getattr(model.model.llm, "folded_layers", 0)),
                       # the last decoder layer's MLP runs on the rows the filtered CE and the <SEG> gather read — reported only when the stack really
                       # pruned (llm.pruned_rows).  Bit-identical losses and gradients on the frozen MoE trunk; with adapters, identical up to fp32 summation
                       # order at lora_dropout 0 and another sample of the same dropout distribution at p > 0 (tests/test_gpu_prune_last_mlp.py); "all"
                       # under MP_PRUNE_LAST_MLP=0
                       "last_layer_mlp_rows": pruned_tflop_per_step(model, cfg, args.lora)[1] or "all",
                       "mask_upsampler": ("fused bf16 kernel, forward + recomputing backward (in the step)" if cfg.fused_bf16_upsampler
                                          else "fp32 tail (6 launches forward, 14 backward)")},
            # algorithmic work per sample: the forward (9.15 TFLOP); with --lora also the decoder's dgrad (8.66: the frozen
            # projections' input gradients + the attention backward; no wgrad for frozen weights)
            # minus what the last layer's MLP skips on rows nothing reads (config.last_layer_mlp_rows)
            "model_tflops_per_gpu": (round(((FWD_TFLOP_PER_SAMPLE + (8.66 if args.lora else 0.0)) * args.batch
                                            - pruned_tflop_per_step(model, cfg, args.lora)[0]) * args.steps / dt, 1) if not args.ep else None),
            "loss_after_warmup": loss0, "loss_last": float(out["loss"].detach()),
            "roofline": roof, "roofline_timed_region": (roof_timed if timer_u is not None else None),
            # data parallel: what RCCL connected, and the gradient bucket (one SUM all-reduce of the flat fp32 gradient on the
            # communication stream) against the tail backward it follows — both per optimizer step, from HIP events on their streams
            "rccl_ranks": rccl_ranks, "dp_bucket": dp_bucket, "mask_tail": tail_obj,
        }
        if ep_obj is not None:
            res["ep"] = ep_obj
        print(f"[bench] gpu leg: {value:.2f} samples/s, {dt / args.steps * 1e3:.1f} ms/step", file=sys.stderr, flush=True)
        if world == 1:
            try:
                model.sync_side_streams()
                torch.cuda.synchronize()
                res["roofline_upsampler"] = upsampler_roofline(device)
            except Exception as e:
                res["roofline_upsampler"] = {"error": f"{type(e).__name__}: {e}"}
        secondary = None
        if world == 1 and not args.lora and not args.ep and not args.no_secondary and (args.experts, args.top_k, args.use_residual) == (2, 1, False):
            # the other BASELINE configurations and evaluate()'s decode, measured in THIS run (round-3 review: they were builder-only figures)
            secondary = {"configs": {}, "decode": {}}
            try:
                model.sync_side_streams(); torch.cuda.synchronize()
                secondary["configs"]["2"] = dict(forward_rate(model, batch, args.batch, seq_len, args.batch),
                                                what="BASELINE configs[2]: MedPLIB-7B-MoE bf16 pixel-grounding forward + SAM-Med2D decoder, batch 8 (the headline model, no_grad)")
                secondary["decode"]["moe"] = decode_rate(model, device)
            except Exception as e:
                secondary["configs"]["2"] = {"error": f"{type(e).__name__}: {e}"}
        if world == 1 and not args.lora and not args.no_lora_line:
            try:
                model.sync_side_streams(); torch.cuda.synchronize()
                res["lora_stage3"] = lora_secondary(args, device, ds_config, synthetic_batch, rank, secondary=secondary)
            except Exception as e:
                res["lora_stage3"] = {"error": f"{type(e).__name__}: {e}"}
        if secondary is not None:
            try:
                # BASELINE configs[4] on ONE rank (its 8-GPU form with the expert all-to-all is `bench.py --gpus 8 --ep 2`)
                from medplib_amd.model.medplib import MedPLIBForCausalLM as _M
                icfg = icl_config(args.layers)
                im = _M(icfg, device=device).train()
                ib, iS = synthetic_icl_batch(icfg, 4, device, seed=42)
                secondary["configs"]["4"] = dict(forward_rate(im, ib, 4, iS, 16),
                                                what="BASELINE configs[4] on one rank: MedPLIB-ICL separate mode forward, 3 in-context (image, mask) pairs + query, "
                                                     "mm_token_compress 576->256, mask encoder, E=2 top-1, batch 4")
                im.sync_side_streams(); torch.cuda.synchronize()
                del im, ib
                torch.cuda.empty_cache()
            except Exception as e:
                secondary["configs"]["4"] = {"error": f"{type(e).__name__}: {e}"}
            res["configs"], res["decode"] = secondary["configs"], secondary["decode"]
        if world == 1 and not args.no_cpu_baseline:
            try:
                # (the parity model is a second set of weights in HBM beside the benchmarked one: 2 x 23 GB of 288)
                res["cpu_baseline"], res["parity"] = cpu_baseline(cfg, device)
            except Exception as e:   # the line is still printed (the GPU number was measured), but a host leg that died is NOT a pass:
                # without it there is no `parity` object and nothing has checked the results — the process exits non-zero (round-4 review)
                res["cpu_baseline"] = {"value": None, "unit": "samples/s", "cores": os.cpu_count(), "kind": "port",
                                       "sample": f"failed: {type(e).__name__}: {e}"}
                print(f"[bench] the oracle / parity leg failed: {type(e).__name__}: {e}", file=sys.stderr, flush=True)
                parity_failed = True
        emit(json.dumps(res))
        if res.get("parity"):
            # the line's own parity object against the bounds the GPU tests assert (oracle/parity.py: check_full_size): a fast line with a
            # wrong result is not a result — the process fails after printing it
            from oracle.parity import check_full_size
            bad = check_full_size(res["parity"], cfg.num_hidden_layers, True)
            if bad:
                print("[bench] PARITY VIOLATED: " + "; ".join(bad), file=sys.stderr, flush=True)
                parity_failed = True
    if world > 1 or force_dist:
        dist.destroy_process_group()
    if parity_failed:
        sys.exit(3)


if __name__ == "__main__":
    main()
