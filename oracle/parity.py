"""ORACLE-SIDE TEST INFRASTRUCTURE ONLY (see oracle/ops.py header): the full-depth, true-dimension parity check.

`full_size_parity(cfg, device)` runs the WHOLE `model_forward` (CLIP tower -> splice -> `cfg.num_hidden_layers` Llama / MoE decoder
layers at 7B dims -> CE; SAM-Med2D encoder -> <SEG> projection -> mask decoder -> postprocess -> 4 mask losses) once on the CPU oracle
(fp32) and once on the HIP path (bf16 trunk, fp32 tail) from the same seeded weights and the same B = 1 batch, and reports how far
apart they are: the 10 losses, the last hidden state, per-layer routing agreement, the thresholded-mask Dice.  Weights: one decoder
layer's seeded weights aliased over all layers on BOTH sides (`init_hf_weights_aliased`), which bounds host memory at true dims.
Gate sampling: either off on both sides (the routing is a function of the gate alone) or DeepSpeed's Random Token Selection with
the SAME uniform draws injected on both sides (`rts_seed`), at B = 8 = the benchmark's T = 5112 tokens, where capacity overflow
decides which tokens are dropped.  Masks are compared where a comparison can fail (`ops.mask_cut_report`).
Called by tests/test_gpu_model.py (8 layers) and by bench.py's cpu_baseline leg (32 layers, un-timed), never by the product."""
import copy
import time

import torch

from . import model as OM
from . import ops as O


MASK_LOGIT_TOL = 0.066         # stated tolerance on the mask logits of the bf16 trunk + bf16 upsampler against the fp32 oracle at full depth =
                               # the measured worst case + 20 % (round 4; it was a round 0.08).  Measured at 32 layers, B = 8: 0.0544 with the
                               # fused bf16 upsampler (the default since round 3), 0.038 with the fp32 tail — the difference is the kernel's
                               # bf16 operand rounding: `src`, W1, W2 and the intermediate a1 are bf16 MFMA operands and the upscaled
                               # embedding is rounded to bf16 before the hypernetwork product, each 2^-9 relative on values of magnitude
                               # 2-3 summed over 32 channels (config.fused_bf16_upsampler=False is the strict fp32 tail).  Pixels whose
                               # reference logit is farther than the MEASURED error from a cut must threshold identically (`flipped <= near_cut`)
                               # Round 5, more samples of the same quantity (the advisor's point: one seed is thin): 0.0534 (driver run r04), 0.0534
                               # (r05c, another box), 0.0489 (32 layers of DISTINCT weights, B = 1, profiles/r05_distinct_parity.json): the bound holds
                               # 19-26 % above every one of them
MASK_LOGIT_TOL_FP32_TAIL = 0.045   # the same quantity with the strict fp32 tail (config.fused_bf16_upsampler=False): everything it holds is the bf16
                               # TRUNK's error carried through an fp32 decoder (measured 0.038 at 32 layers, B = 8).  Asserted beside the fused bound
                               # (tests/test_gpu_model.py::test_full_depth_parity_fp32_tail_binds_the_trunk) so that a trunk regression cannot hide in
                               # the fused kernel's larger allowance (round-5 review, weak 1b)
HIDDEN_P999_ALL_ROWS = 0.05    # the 99.9th-percentile element error of the last hidden state over ALL rows, relative to the largest reference entry.
HIDDEN_BAD_ROW = 0.1           # a row is "bad" when its worst element is off by more than this (same scale).  A token that picked the other
                               # expert in some layer is a different computation from there on: its row differs by O(its own magnitude) and
                               # only mixes back through attention, so no tight bound holds for THAT row (measured worst element 0.128 at 32
                               # layers with 4.5 % of the rows flipped somewhere, 0.49 at 8 layers) — but flips are the ONLY licence for a bad
                               # row.  Three bounds that can fail (round-4 review; the former all-rows bound of 1.0 caught NaN and nothing else):
                               # rows agreeing in every layer: worst element <= 0.1 (measured 0.035-0.076); all rows: 99.9th percentile element
                               # <= 0.05; bad rows <= 2 x the flipped tokens summed over the layers (a flip damages its own row and, through
                               # the capacity boundary it moves, at most one more)

def check_full_size(r, layers, moe):
    """-> list of violated bounds (empty = parity holds) for a full_size_parity() result; ONE statement of the bounds for
    tests/test_gpu_model.py (`_assert_full_size`) and for bench.py, whose line fails (non-zero exit) when its own `parity` object
    violates them (round-3 review, item 8)."""
    bad = []

    def need(ok, what):
        if not ok:
            bad.append(what)
    need(r["max_abs_dloss_over_10"] < 5e-2, f"losses: max |d| over the 10 = {r['max_abs_dloss_over_10']:.4g} >= 5e-2")
    need(r["hidden_rel_err_agreeing_rows"] < 0.1, f"hidden (rows agreeing in every layer): {r['hidden_rel_err_agreeing_rows']:.4g} >= 0.1")
    need(r["hidden_rel_err"] == r["hidden_rel_err"] and r["hidden_rel_err"] < float("inf"), "hidden: not finite")
    # rows that agree in every layer: always; ALL rows: while at most 5 % of the rows flipped somewhere (every standing configuration: 0.6-4.5 %).
    # With distinct random gates in 32 layers a sixth of the rows flips at least once (scripts/r05_distinct_parity.py: 16 %) and the percentile
    # lands inside the flipped rows, which no bound can hold — there the agreeing rows and the bad-row count carry the check
    need(r["hidden_p999_rel_err_agreeing_rows"] <= HIDDEN_P999_ALL_ROWS,
         f"hidden (agreeing rows, 99.9th percentile element): {r['hidden_p999_rel_err_agreeing_rows']:.4g} > {HIDDEN_P999_ALL_ROWS}")
    few_flips = r["rows_agreeing_in_every_layer"] >= 0.95
    # an ABSOLUTE floor, so that a regression which flips many tokens cannot loosen its own acceptance (the all-rows bounds below switch off and
    # the bad-row allowance grows with the flips): every standing (aliased-weights) configuration keeps >= 95 % of the rows in agreement in every
    # layer (measured 95.5-99.4 %), and the distinct-weights run, where independent random gates flip a sixth of the rows over 32 layers, >= 75 %
    floor = 0.75 if r.get("distinct_weights") else 0.93               # 0.93: two points under the measured worst (0.955 at 32 layers, B = 8)
    need(r["rows_agreeing_in_every_layer"] >= floor,
         f"only {r['rows_agreeing_in_every_layer']:.4f} of the rows agree with the oracle's routing in every layer (floor {floor})")
    need(r["hidden_bad_rows"] <= (0.25 if r.get("distinct_weights") else 0.05) * r.get("rows_total", float("inf")),
         f"hidden: {r['hidden_bad_rows']} bad rows of {r.get('rows_total')}: above the absolute cap")
    need(not few_flips or r["hidden_p999_rel_err"] <= HIDDEN_P999_ALL_ROWS,
         f"hidden (all rows, 99.9th percentile element): {r['hidden_p999_rel_err']:.4g} > {HIDDEN_P999_ALL_ROWS}")
    need(r["hidden_bad_rows"] <= 2 * r["flipped_tokens_total"],
         f"hidden: {r['hidden_bad_rows']} rows off by more than {HIDDEN_BAD_ROW} with only {r['flipped_tokens_total']} flipped tokens over all layers")
    # mean element error: a bf16 residual stream takes about four roundings of 2^-9 per layer (attention output, its residual add, MLP output, its
    # residual add) that accumulate as a random walk: sqrt(4 L) 2^-9 = 0.0156 at 8 layers (= the 2^-6 this bound has always been), 0.0221 at 32.
    # Measured at 32 layers: 0.0127 with aliased weights (B = 8), 0.0174 over the agreeing rows with distinct weights (B = 1)
    mean_bound = max(2 ** -6, (4 * layers) ** 0.5 * 2 ** -9)
    need(r["hidden_mean_rel_err_agreeing_rows"] < mean_bound,
         f"hidden mean error over the agreeing rows {r['hidden_mean_rel_err_agreeing_rows']:.4g} >= {mean_bound:.4g}")
    need(r["rows_agreeing_in_every_layer"] < 0.95 or r["hidden_mean_rel_err"] < mean_bound, f"hidden mean error {r['hidden_mean_rel_err']:.4g} >= {mean_bound:.4g}")
    mk = r["mask"]
    tol = MASK_LOGIT_TOL if r.get("fused_bf16_upsampler", True) else MASK_LOGIT_TOL_FP32_TAIL
    need(mk["max_abs_dlogit"] <= tol, f"mask logits: max |d| {mk['max_abs_dlogit']:.4g} > {tol}")
    for c in ("cut_ref", "cut_zero"):
        need(mk[c]["flipped_le_near_cut_every_mask"], f"{c}: a mask has more flipped pixels than pixels inside the error band")
        need(mk[c]["max_abs_ddice"] <= 1e-3, f"{c}: |dDice| {mk[c]['max_abs_ddice']:.4g} > 1e-3")
    if moe:
        need(len(r["routing_agreement_per_layer"]) == layers, "routing report does not cover every layer")
        need(r["routing_agreement_min"] is not None and r["routing_agreement_min"] >= 0.97, f"routing agreement min {r['routing_agreement_min']} < 0.97")
        ll = r.get("routing_layer_local")
        need(ll is not None and ll["layers"] == layers, "layer-local routing report missing or incomplete")
        if ll is not None:
            # identical inputs, identical rounding points: only a tie within rounding may flip (measured margins of such ties: <= 1e-4 on logits of O(1))
            need(ll["agreement_min"] >= 0.9995, f"layer-local routing agreement {ll['agreement_min']} < 0.9995")
            need(ll["max_flip_margin"] <= 2e-3, f"a token flipped against a logit margin of {ll['max_flip_margin']:.3g} > 2e-3 on its own layer's input")
        rt = r["routing"]
        need(rt["kept_set_equals_deepspeed_rule_every_layer"] and rt["slots_equal_deepspeed_rule_every_layer"], "kept set / slots differ from DeepSpeed's rule")
        need(rt["counts_equal_own_choices_every_layer"] and rt["kept_sets_bit_equal_where_choices_identical"], "expert counts / kept sets")
        # against the oracle's own run: a token that flipped moves the capacity boundary of its old and its new expert by one each
        need(all(d <= 2 * f for d, f in zip(rt["kept_state_differs_on_agreeing_rows_per_layer"], rt["flipped_tokens_per_layer"])),
             "kept state differs on more agreeing rows than 2 x flipped tokens")
    return bad


def _to_dev(batch, device):
    gb = {k: (v.to(device) if torch.is_tensor(v) else v) for k, v in batch.items()}
    for k in ("masks_list", "images_clip", "mask_images"):
        if isinstance(batch.get(k), (list, tuple)):
            gb[k] = [x.to(device) for x in batch[k]]
    return gb


def routing_report(coll, routing, T, capacity, rts):
    """Per MoE layer: the HIP path's routing against (i) the oracle's routing of the same layer and (ii) DeepSpeed's selection rule
    applied on the host to the HIP path's OWN expert choices with the same injected uniforms (`llm.top1_capacity_selection`) — the
    second is exact by construction (no bf16 noise enters it): kept / dropped sets, slots and counts must be bit-equal in every
    layer, at the benchmark's token count, whether or not an upstream gate probability flipped.  coll: oracle (idx, slot, counts)
    per layer; routing: HIP (expert, slot, counts) per layer."""
    import torch.nn.functional as F
    from . import llm
    per, same_all = [], torch.ones(T, dtype=torch.bool)
    for li, ((e_ref, s_ref, c_ref), r) in enumerate(zip(coll, routing)):
        if isinstance(e_ref, (tuple, list)):
            # top-2 layers: first and second choice per token must agree (in order); the capacity rule below restates top-1 selection only.
            # oracle: (idx1, idx2), each [T]; HIP: entry arrays of length 2T (first choices, then second choices)
            e_ref = torch.stack([e_ref[0][:T], e_ref[1][:T]], 1).long()
            eh = r[0].cpu().long()
            e_hip2 = torch.stack([eh[:T], eh[eh.numel() // 2:eh.numel() // 2 + T]], 1)
            agree = (e_hip2 == e_ref).all(1)
            same_all &= agree
            sh = r[1].cpu().long()
            kept_hip2 = torch.stack([sh[:T], sh[sh.numel() // 2:sh.numel() // 2 + T]], 1) >= 0
            kept_ref2 = torch.stack([s_ref[0][:T], s_ref[1][:T]], 1) >= 0
            per.append({"layer": li, "expert_agreement": float(agree.float().mean()), "flipped_tokens": int((~agree).sum()),
                        # entries (token, choice) dropped for capacity on either side; on agreeing tokens the kept state may differ only
                        # where a flipped token moved an expert's capacity boundary (each flip touches up to four queues)
                        "dropped_entries_hip": int((~kept_hip2).sum()), "dropped_entries_oracle": int((~kept_ref2).sum()),
                        "kept_state_differs_on_agreeing_rows": int(((kept_hip2 != kept_ref2).any(1) & agree).sum()),
                        "first_choice_agreement": float((e_hip2[:, 0] == e_ref[:T, 0]).float().mean()),
                        "counts_equal_oracle": bool(torch.equal(r[2].cpu().long().view(-1)[:c_ref.numel()], c_ref.view(-1).long()))})
            continue
        e_hip, s_hip, c_hip = r[0].cpu().long()[:T], r[1].cpu().long()[:T], r[2].cpu().long()
        E = int(c_ref.numel())
        agree = e_hip == e_ref[:T]
        same_all &= agree
        kept_hip, kept_ref = s_hip >= 0, s_ref[:T] >= 0
        _, slot_rule, kept_rule = llm.top1_capacity_selection(F.one_hot(e_hip, num_classes=E), capacity, None if rts is None else rts[li])
        slot_rule = torch.where(kept_rule, slot_rule, torch.full_like(slot_rule, -1))
        per.append({"layer": li, "expert_agreement": float(agree.float().mean()), "flipped_tokens": int((~agree).sum()),
                    "dropped_hip": int((~kept_hip).sum()), "dropped_oracle": int((~kept_ref).sum()),
                    # against the oracle's own run: a flipped token moves the capacity boundary of both experts by one
                    "kept_state_differs_on_agreeing_rows": int((kept_hip != kept_ref)[agree].sum()),
                    # against the rule on the HIP path's own choices: exact
                    "kept_set_equals_rule": bool(torch.equal(kept_hip, kept_rule)), "slots_equal_rule": bool(torch.equal(s_hip, slot_rule)),
                    "counts_equal_own_choices": bool(torch.equal(c_hip[:E], torch.bincount(e_hip, minlength=E)))})
    return per, same_all


def layer_local_routing(routing, W, cfg, top_k, gate_inputs=None):
    """The routing of every MoE layer recomputed on the host from the HIP path's OWN input to that layer's gate (the residual stream in
    front of the post-attention norm, which the model hands over in collect mode): fp32 RMSNorm, rounded to bf16 as the kernel's normed row
    is, fp32 logits, argmax.  With identical inputs and identical rounding points only the summation order differs, so a token may pick
    another expert only when its two best logits are within rounding of each other — upstream bf16 noise, which makes 1 % of the tokens of
    the full-depth comparison against the fp32 oracle flip, cannot enter here: a routing error of any rate shows as flips with a real margin.
    -> {"agreement_min", "flips_total", "max_flip_margin", "tokens"} (round-4 review, parity item c)."""
    agree, flips, worst, T = [], 0, 0.0, 0
    moe_ids = sorted(cfg.moe_layer_set())
    if not gate_inputs or len(gate_inputs) != len(routing):
        return None
    for li, r, xg in zip(moe_ids, routing, gate_inputs):
        if xg is None:
            return None
        x = xg.float().cpu()
        T = x.shape[0]
        p = f"model.layers.{li}."
        # the kernel's (= HF LlamaRMSNorm's) two rounding points: the normalised value is cast to bf16 BEFORE the weight multiplies it, the
        # product is cast again (csrc/ce_moe.hip rmsnorm_gate_kernel; one rounding here instead made 19 of 20 448 tokens flip with margins up
        # to 0.017 in the first distinct-weights run, profiles/r05_distinct_parity_first.json: exactly the tie rate one bf16 rounding predicts)
        rs = torch.rsqrt(x.pow(2).mean(-1, keepdim=True) + cfg.rms_norm_eps)
        h = (W[p + "post_attention_layernorm.weight"].float() * (x * rs).to(torch.bfloat16).float()).to(torch.bfloat16).float()
        logits = h @ W[p + "mlp.deepspeed_moe.gate.wg.weight"].float().t()
        ref = logits.argmax(1)
        hip = r[0].cpu().long().view(-1)[:T]                       # first choices (top-2 entry arrays start with them)
        diff = ref != hip
        top2 = logits.topk(2, dim=1).values
        margin = (top2[:, 0] - top2[:, 1])
        agree.append(1.0 - float(diff.float().mean()))
        flips += int(diff.sum())
        if diff.any():
            worst = max(worst, float(margin[diff].max()))
    return {"agreement_min": min(agree) if agree else None, "flips_total": flips, "max_flip_margin": worst, "tokens": T, "layers": len(agree)}


def full_size_parity(cfg, device, seed=0, batch_seed=42, H=336, Wd=336, cpu_threads=None, time_oracle=None, icl_ctx=0, B=1,
                     rts_seed=None, capacity_factor=None, time_threads=None, prompt_len=64, ragged=False, time_forward=None, distinct_weights=False):
    """-> dict of plain numbers.  `time_oracle=(warmup, timed)`: also time the oracle's B = 1 training step (forward + backward
    through the trainable tail) that many times and return the per-step seconds (bench.py's cpu_baseline).
    B: samples in the compared batch (8 = the benchmark's per-GPU batch, T = 5112).  rts_seed: DeepSpeed's Random Token Selection
    ON with the same uniform draws injected on both sides (None: sampling off, first-come selection).  capacity_factor: override
    (1.0 = the reference driver's argparse default, train_ds_medplib.py:127: the larger expert overflows in every layer, so the
    RTS selection decides which tokens are dropped; 1.5 = scripts/train_stage4.sh, where balanced gates never overflow)."""
    from medplib_amd.model.medplib import LISAForCausalLM, MedPLIBForCausalLM
    if cpu_threads:
        torch.set_num_threads(cpu_threads)
    cfg = copy.deepcopy(cfg)
    cfg.moe_gate_sampling = False
    if capacity_factor is not None:
        cfg.capacity_factor = capacity_factor
    # distinct_weights: every decoder layer gets its OWN seeded weights (a per-layer weight-indexing error at depth cannot hide behind
    # aliasing; scripts/r05_distinct_parity.py) — stored as the bf16 values they are and upcast one matrix at a time (model.UpcastDict)
    W = OM.init_hf_weights(cfg, seed=seed, store_bf16=True) if distinct_weights else OM.init_hf_weights_aliased(cfg, seed=seed)

    def make(B_, bseed):
        if icl_ctx:                               # BASELINE config 5 shape: icl_ctx in-context (image, mask) pairs + the query, separate mode
            b = OM.make_batch_icl(cfg, B_, n_ctx=icl_ctx, H=H, Wd=Wd, seed=bseed, mask_size=cfg.clip_image_size)
            b["images_clip"] = [x.to(torch.bfloat16).float() for x in b["images_clip"]]
        else:
            b = OM.make_batch(cfg, B_, L=prompt_len, H=H, Wd=Wd, seed=bseed, ragged=ragged)
            b["images_clip"] = b["images_clip"].to(torch.bfloat16).float()
        b["images"] = b["images"].to(torch.bfloat16).float()
        return b
    times, ftimes = [], []
    if time_forward:
        if time_threads:
            torch.set_num_threads(time_threads)
        b1 = make(1, batch_seed)
        with torch.no_grad():
            for _ in range(time_forward[0] + time_forward[1]):
                t0 = time.time()
                OM.model_forward(b1, W, cfg, training=True)
                ftimes.append(time.time() - t0)
        ftimes = ftimes[time_forward[0]:]
        if cpu_threads:
            torch.set_num_threads(cpu_threads)
    if time_oracle:
        if time_threads:
            torch.set_num_threads(time_threads)
        b1 = make(1, batch_seed)
        train = [k for k in W if k.startswith("model.visual_model.mask_decoder.") or k.startswith("model.text_hidden_fcs.")]
        Wt = dict(W)
        for k in train:
            Wt[k] = W[k].clone().requires_grad_()
        for _ in range(time_oracle[0] + time_oracle[1]):
            for k in train:
                Wt[k].grad = None
            t0 = time.time()
            out = OM.model_forward(b1, Wt, cfg, training=True)
            out["loss"].backward()
            times.append(time.time() - t0)
        times = times[time_oracle[0]:]
        del Wt, out
        if cpu_threads:
            torch.set_num_threads(cpu_threads)
    batch = make(B, batch_seed)
    rts = None
    if rts_seed is not None and cfg.moe_enable:
        g = torch.Generator().manual_seed(rts_seed)
        S_ = batch["input_ids"].shape[1] - 1 + cfg.image_token_len if not icl_ctx else None     # the padded length (ragged rows are padded to it)
        assert S_ is not None, "injected draws need the spliced length up front (single-image layout)"
        rts = {i: torch.rand(B * S_, cfg.num_experts, generator=g) for i in sorted(cfg.moe_layer_set())}
    coll = []
    t0 = time.time()
    with torch.no_grad():
        ref, inter = OM.model_forward(batch, W, cfg, training=True, return_intermediates=True, collect=coll, rts=rts)
    t_oracle = time.time() - t0

    cls = MedPLIBForCausalLM if cfg.moe_enable else LISAForCausalLM
    m = cls(cfg, device=device).train()
    m.load_hf_state_dict(W)
    m.capture_intermediates = True
    if rts is not None:
        moe_ids = sorted(cfg.moe_layer_set())
        rts_dev = {i: rts[i].to(device) for i in moe_ids}
        m.model.llm.rts_uniform_provider = lambda i, T_, E_: rts_dev[i]
        rts_list = [rts[i] for i in moe_ids]
    else:
        rts_list = None
    gb = _to_dev(batch, device)
    with torch.no_grad():
        out = m(**gb)
        folded_layers = int(getattr(m.model.llm, "folded_layers", 0))       # decoder layers that ran config.fold_input_norm's kernels in THIS pass
        losses_gpu = {k: float(out[k]) for k in O.LOSS_KEYS}
        cap = m.captured
        hid = cap["last_hidden"].float().cpu()
        routing = cap.get("routing") or []
        T = hid.shape[0] * hid.shape[1]
        # the per-layer routing report restates top-1 selection; top-2 layers are compared through their outputs only
        per_layer, same = routing_report(coll, routing, T, m.model.llm.capacity(T), rts_list) if routing else ([], torch.ones(T, dtype=torch.bool))
        agree = [p["expert_agreement"] for p in per_layer]
        local = layer_local_routing(routing, W, cfg, cfg.top_k_experts, getattr(m.model.llm, "last_gate_inputs", None)) if routing else None
        masks = m(**dict(gb, inference=True))["pred_masks"]
    losses_cpu = {k: float(ref[k]) for k in O.LOSS_KEYS}
    # the masks, where a comparison can fail (oracle/ops.py: mask_cut_report): per mask at the reference's cut and at logit 0
    reports = [O.mask_cut_report(masks[i][0].float().cpu(), inter["pred_masks"][i][0], batch["masks_list"][i]) for i in range(len(masks))]
    def agg(cut, key, f=max):
        return f(r[cut][key] for r in reports)
    mask_summary = {"masks": len(reports), "pixels_per_mask": reports[0]["pixels"],
                    "max_abs_dlogit": max(r["max_abs_dlogit"] for r in reports), "mean_abs_dlogit": sum(r["mean_abs_dlogit"] for r in reports) / len(reports),
                    "logit_tolerance": MASK_LOGIT_TOL}
    for cut in ("cut_ref", "cut_zero"):
        mask_summary[cut] = {"logit_cut": reports[0][cut]["logit_cut"],
                             "pos_frac_ref_min_max": [agg(cut, "pos_frac_ref", min), agg(cut, "pos_frac_ref", max)],
                             "pos_frac_pred_min_max": [agg(cut, "pos_frac_pred", min), agg(cut, "pos_frac_pred", max)],
                             "flipped_total": sum(r[cut]["flipped"] for r in reports), "near_cut_total": sum(r[cut]["near_cut"] for r in reports),
                             "flipped_le_near_cut_every_mask": all(r[cut]["flipped"] <= r[cut]["near_cut"] for r in reports),
                             "max_abs_ddice": agg(cut, "abs_ddice"), "dice_ref": [round(r[cut]["dice_ref"], 6) for r in reports],
                             "dice_pred": [round(r[cut]["dice_pred"], 6) for r in reports]}
    href = inter["hidden"]
    d = hid.shape[-1]
    res = {"layers": cfg.num_hidden_layers, "moe": bool(cfg.moe_enable), "batch": B, "seq_len": int(href.shape[1]), "tokens": int(T),
           "capacity_factor": cfg.capacity_factor, "capacity": int(m.model.llm.capacity(T)) if cfg.moe_enable else None,
           "gate_sampling": ("RTS on, identical uniforms injected on both sides (seed %d)" % rts_seed) if rts is not None else "off",
           "abs_dloss": abs(losses_gpu["loss"] - losses_cpu["loss"]),
           "max_abs_dloss_over_10": max(abs(losses_gpu[k] - losses_cpu[k]) for k in O.LOSS_KEYS),
           "loss_gpu": losses_gpu["loss"], "loss_cpu": losses_cpu["loss"], "ce_gpu": losses_gpu["ce_loss"], "ce_cpu": losses_cpu["ce_loss"],
           "mask_loss_gpu": losses_gpu["mask_loss"], "mask_loss_cpu": losses_cpu["mask_loss"],
           "hidden_rel_err": float((hid - href).abs().max() / href.abs().max()),
           "hidden_mean_rel_err": float((hid - href).abs().mean() / href.abs().mean()),
           "hidden_p999_rel_err": float(torch.kthvalue((hid - href).abs().flatten(), max(1, int(0.999 * hid.numel()))).values / href.abs().max()),
           "hidden_bad_rows": int(((hid - href).abs().view(-1, hid.shape[-1]).max(1).values > HIDDEN_BAD_ROW * href.abs().max()).sum()),
           "hidden_p999_rel_err_agreeing_rows": float(torch.kthvalue((hid.view(-1, d)[same] - href.view(-1, d)[same]).abs().flatten(),
                                                                     max(1, int(0.999 * int(same.sum()) * d))).values / href.abs().max()) if same.any() else 0.0,
           "hidden_mean_rel_err_agreeing_rows": float((hid.view(-1, d)[same] - href.view(-1, d)[same]).abs().mean() / href.view(-1, d)[same].abs().mean()) if same.any() else 0.0,
           "flipped_tokens_total": int(sum(p["flipped_tokens"] for p in per_layer)),
           # a token that picked the other expert somewhere is a different computation from there on: bound the rest
           "hidden_rel_err_agreeing_rows": float((hid.view(-1, d)[same] - href.view(-1, d)[same]).abs().max() / href.abs().max()),
           "rows_agreeing_in_every_layer": float(same.float().mean()),
           "rows_total": int(same.numel()),
           "mask": mask_summary,
           # kept for continuity with earlier rounds' lines (Dice of mask 0 at the reference cut; see `mask` for what bites)
           "dice_gpu": reports[0]["cut_ref"]["dice_pred"], "dice_cpu": reports[0]["cut_ref"]["dice_ref"],
           "abs_ddice": max(mask_summary["cut_ref"]["max_abs_ddice"], mask_summary["cut_zero"]["max_abs_ddice"]),
           "mask_logit_max_abs_err": mask_summary["max_abs_dlogit"],
           "routing_agreement_min": min(agree) if agree else None,
           "routing_agreement_mean": (sum(agree) / len(agree)) if agree else None,
           "routing_agreement_per_layer": [round(a, 4) for a in agree],
           "routing_layer_local": local,
           "oracle_forward_seconds": round(t_oracle, 2),
           "distinct_weights": bool(distinct_weights),
           "folded_layers": folded_layers,
           "fused_bf16_upsampler": bool(getattr(cfg, "fused_bf16_upsampler", False)),
           "weights": ("DISTINCT seeded weights in every decoder layer, both sides" if distinct_weights
                       else "one decoder layer's seeded weights aliased over all layers, both sides")}
    if per_layer and "first_choice_agreement" in per_layer[0]:        # top-2 layers
        res["routing"] = {"top_k": 2, "flipped_tokens_per_layer": [p["flipped_tokens"] for p in per_layer],
                          "dropped_entries_hip_per_layer": [p["dropped_entries_hip"] for p in per_layer],
                          "dropped_entries_oracle_per_layer": [p["dropped_entries_oracle"] for p in per_layer],
                          "kept_state_differs_on_agreeing_rows_per_layer": [p["kept_state_differs_on_agreeing_rows"] for p in per_layer],
                          "first_choice_agreement_per_layer": [round(p["first_choice_agreement"], 4) for p in per_layer],
                          "counts_equal_oracle_where_choices_identical": all(p["counts_equal_oracle"] for p in per_layer if p["flipped_tokens"] == 0)}
    elif per_layer:
        res["routing"] = {"dropped_hip_per_layer": [p["dropped_hip"] for p in per_layer],
                          "dropped_oracle_per_layer": [p["dropped_oracle"] for p in per_layer],
                          "flipped_tokens_per_layer": [p["flipped_tokens"] for p in per_layer],
                          "kept_state_differs_on_agreeing_rows_per_layer": [p["kept_state_differs_on_agreeing_rows"] for p in per_layer],
                          "layers_with_identical_choices": sum(p["flipped_tokens"] == 0 for p in per_layer),
                          "kept_sets_bit_equal_where_choices_identical": all(p["kept_state_differs_on_agreeing_rows"] == 0 and p["dropped_hip"] == p["dropped_oracle"]
                                                                             for p in per_layer if p["flipped_tokens"] == 0),
                          "kept_set_equals_deepspeed_rule_every_layer": all(p["kept_set_equals_rule"] for p in per_layer),
                          "slots_equal_deepspeed_rule_every_layer": all(p["slots_equal_rule"] for p in per_layer),
                          "counts_equal_own_choices_every_layer": all(p["counts_equal_own_choices"] for p in per_layer)}
    if times:
        res["oracle_step_seconds"] = times
    if ftimes:
        res["oracle_forward_b1_seconds"] = ftimes
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

