"""ORACLE-SIDE TEST INFRASTRUCTURE ONLY (see oracle/ops.py header): the full-depth, true-dimension parity check.

`full_size_parity(cfg, device)` runs the WHOLE `model_forward` (CLIP tower -> splice -> `cfg.num_hidden_layers` Llama / MoE decoder
layers at 7B dims -> CE; SAM-Med2D encoder -> <SEG> projection -> mask decoder -> postprocess -> 4 mask losses) once on the CPU oracle
(fp32) and once on the HIP path (bf16 trunk, fp32 tail) from the same seeded weights and the same B = 1 batch, and reports how far
apart they are: the 10 losses, the last hidden state, per-layer routing agreement, the thresholded-mask Dice.  Weights: one decoder
layer's seeded weights aliased over all layers on BOTH sides (`init_hf_weights_aliased`), which bounds host memory at true dims.
Gate sampling (DeepSpeed's RTS draws) is off on both sides so the routing is a function of the gate alone.
Called by tests/test_gpu_model.py (8 layers) and by bench.py's cpu_baseline leg (32 layers, un-timed), never by the product."""
import copy
import time

import torch

from . import model as OM
from . import ops as O


def full_size_parity(cfg, device, seed=0, batch_seed=42, H=336, Wd=336, cpu_threads=None, time_oracle=None, icl_ctx=0):
    """-> dict of plain numbers.  `time_oracle=(warmup, timed)`: also time the oracle's B = 1 training step (forward + backward
    through the trainable tail) that many times and return the per-step seconds (bench.py's cpu_baseline)."""
    from medplib_amd.model.medplib import LISAForCausalLM, MedPLIBForCausalLM
    if cpu_threads:
        torch.set_num_threads(cpu_threads)
    cfg = copy.deepcopy(cfg)
    cfg.moe_gate_sampling = False
    W = OM.init_hf_weights_aliased(cfg, seed=seed)
    if icl_ctx:                                   # BASELINE config 5 shape: icl_ctx in-context (image, mask) pairs + the query, separate mode
        batch = OM.make_batch_icl(cfg, 1, n_ctx=icl_ctx, H=H, Wd=Wd, seed=batch_seed, mask_size=cfg.clip_image_size)
        batch["images_clip"] = [x.to(torch.bfloat16).float() for x in batch["images_clip"]]
    else:
        batch = OM.make_batch(cfg, 1, L=64, H=H, Wd=Wd, seed=batch_seed)
        batch["images_clip"] = batch["images_clip"].to(torch.bfloat16).float()
    batch["images"] = batch["images"].to(torch.bfloat16).float()
    train = [k for k in W if k.startswith("model.visual_model.mask_decoder.") or k.startswith("model.text_hidden_fcs.")]
    Wt = dict(W)
    for k in train:
        Wt[k] = W[k].clone().requires_grad_()
    times = []
    if time_oracle:
        for _ in range(time_oracle[0] + time_oracle[1]):
            for k in train:
                Wt[k].grad = None
            t0 = time.time()
            out = OM.model_forward(batch, Wt, cfg, training=True)
            out["loss"].backward()
            times.append(time.time() - t0)
        times = times[time_oracle[0]:]
    coll = []
    with torch.no_grad():
        ref, inter = OM.model_forward(batch, W, cfg, training=True, return_intermediates=True, collect=coll)

    cls = MedPLIBForCausalLM if cfg.moe_enable else LISAForCausalLM
    m = cls(cfg, device=device).train()
    m.load_hf_state_dict(W)
    m.capture_intermediates = True
    gb = {k: (v.to(device) if torch.is_tensor(v) else v) for k, v in batch.items()}
    for k in ("masks_list", "images_clip", "mask_images"):
        if isinstance(batch.get(k), (list, tuple)):
            gb[k] = [x.to(device) for x in batch[k]]
    with torch.no_grad():
        out = m(**gb)
        losses_gpu = {k: float(out[k]) for k in O.LOSS_KEYS}
        cap = m.captured
        hid = cap["last_hidden"].float().cpu()
        routing = cap.get("routing") or []
        agree = []
        same = torch.ones(hid.shape[0] * hid.shape[1], dtype=torch.bool)
        for (e_ref, _, _), r in zip(coll, routing):
            eq = r[0].cpu().long()[: same.numel()] == e_ref[: same.numel()]
            agree.append(float(eq.float().mean()))
            same &= eq
        masks = m(**dict(gb, inference=True))["pred_masks"]
    losses_cpu = {k: float(ref[k]) for k in O.LOSS_KEYS}
    _, _, _, dice_cpu = O.threshold_iou(inter["pred_masks"][0][0], batch["masks_list"][0])
    _, _, _, dice_gpu = O.threshold_iou(masks[0][0].float().cpu(), batch["masks_list"][0])
    href = inter["hidden"]
    res = {"layers": cfg.num_hidden_layers, "moe": bool(cfg.moe_enable), "batch": 1, "seq_len": int(href.shape[1]),
           "abs_dloss": abs(losses_gpu["loss"] - losses_cpu["loss"]),
           "max_abs_dloss_over_10": max(abs(losses_gpu[k] - losses_cpu[k]) for k in O.LOSS_KEYS),
           "loss_gpu": losses_gpu["loss"], "loss_cpu": losses_cpu["loss"], "ce_gpu": losses_gpu["ce_loss"], "ce_cpu": losses_cpu["ce_loss"],
           "mask_loss_gpu": losses_gpu["mask_loss"], "mask_loss_cpu": losses_cpu["mask_loss"],
           "hidden_rel_err": float((hid - href).abs().max() / href.abs().max()),
           "hidden_mean_rel_err": float((hid - href).abs().mean() / href.abs().mean()),
           # a token that picked the other expert somewhere is a different computation from there on: bound the rest
           "hidden_rel_err_agreeing_rows": float((hid.view(-1, hid.shape[-1])[same] - href.view(-1, href.shape[-1])[same]).abs().max()
                                                 / href.abs().max()),
           "rows_agreeing_in_every_layer": float(same.float().mean()),
           "dice_gpu": dice_gpu, "dice_cpu": dice_cpu, "abs_ddice": abs(dice_gpu - dice_cpu),
           "mask_logit_max_abs_err": float((masks[0][0].float().cpu() - inter["pred_masks"][0][0]).abs().max()),
           "routing_agreement_min": min(agree) if agree else None,
           "routing_agreement_mean": (sum(agree) / len(agree)) if agree else None,
           "routing_agreement_per_layer": [round(a, 4) for a in agree],
           "weights": "one decoder layer's seeded weights aliased over all layers, both sides; gate sampling off"}
    if times:
        res["oracle_step_seconds"] = times
    del m
    torch.cuda.empty_cache()
    return res


def lora_grad_parity(cfg, device, r=8, alpha=16, targets="gate_proj,up_proj,down_proj", seed=0, batch_seed=42, cpu_threads=None,
                     sft_modules="mask_decoder,text_hidden_fcs"):
    """LoRA training step at the TRUE layer dimensions (dense decoder of cfg.num_hidden_layers layers, B = 1, S = 639): every adapter
    gradient of the HIP path (the whole decoder backward: attention backward, RMSNorm / SwiGLU backward, dgrad GEMMs on 320- / 256-row
    tiles, the fused adapter branch, MFMA weight gradients) against torch autograd of the oracle with the same adapters in fp32
    (dropout 0: the two sides cannot share a mask stream).  -> {"worst_rel": max over adapters of max|g_hip - g_ref| / max|g_ref|, ...}."""
    from medplib_amd import engine
    from medplib_amd.model.medplib import LISAForCausalLM, MedPLIBForCausalLM
    if cpu_threads:
        torch.set_num_threads(cpu_threads)
    cfg = copy.deepcopy(cfg)
    cfg.moe_gate_sampling = False                  # MoE: routing is a function of the gate alone on both sides (as in full_size_parity)
    W = OM.init_hf_weights_aliased(cfg, seed=seed)
    m = (MedPLIBForCausalLM if cfg.moe_enable else LISAForCausalLM)(cfg, device=device).train()
    m.load_hf_state_dict(W)
    lora = m.enable_lora(lora_r=r, lora_alpha=alpha, lora_dropout=0.0, lora_target_modules=targets, sft_modules=sft_modules)
    g = torch.Generator().manual_seed(seed + 31)
    Wl = dict(W)
    Wl["lora_scaling"] = alpha / r
    for n, p_ in zip(lora.names, lora.params):
        if "lora_" in n:
            v = (torch.randn(p_.shape, generator=g) * (0.02 if "lora_A" in n else 0.01)).to(torch.bfloat16).float()
            p_.data.copy_(v.to(device))
        else:                                      # e.g. a trainable gate `wg`: the checkpoint's own values
            v = p_.detach().float().cpu()
        Wl[n] = v.clone().requires_grad_(True)
    batch = OM.make_batch(cfg, 1, L=64, H=336, Wd=336, seed=batch_seed)
    batch["images"] = batch["images"].to(torch.bfloat16).float()
    batch["images_clip"] = batch["images_clip"].to(torch.bfloat16).float()
    t0 = time.time()
    ref = OM.model_forward(batch, Wl, cfg, training=True, llm_grad=True)
    ref["loss"].backward()
    t_ref = time.time() - t0
    eng, _, _, _ = engine.initialize(model=m, model_parameters=m.trainable_parameters(),
                                     config={"optimizer": {"params": {"lr": 1e-4, "betas": (0.9, 0.95)}}, "gradient_clipping": 1.0})
    gb = {k: (v.to(device) if torch.is_tensor(v) else v) for k, v in batch.items()}
    gb["masks_list"] = [x.to(device) for x in batch["masks_list"]]
    out = eng(**gb)
    losses = {k: (float(out[k].detach()), float(ref[k])) for k in O.LOSS_KEYS}
    eng.backward(out["loss"])
    torch.cuda.synchronize()
    per, worst = {}, 0.0
    for n, p_ in zip(lora.names, lora.params):
        want = Wl[n].grad
        rel = (p_.grad.float().cpu() - want).abs().max().item() / (want.abs().max().item() + 1e-30)
        per[n] = rel
        worst = max(worst, rel)
    return {"layers": cfg.num_hidden_layers, "adapters": len(per), "worst_rel": worst, "per_param": per, "losses_hip_vs_oracle": losses,
            "max_abs_dloss": max(abs(a - b) for a, b in losses.values()), "oracle_seconds": t_ref,
            "grad_absmax_min": min(Wl[n].grad.abs().max().item() for n in lora.names),
            "zero_gradients": [n for n in lora.names if Wl[n].grad.abs().max().item() == 0.0]}

