"""CPU: the oracle restatement reproduces the golden vectors generated from the reference (oracle/make_golden.py)."""
import os

import numpy as np
import torch

from oracle import ops, sam


def test_sam_oracle_matches_reference_golden(golden_dir):
    g = np.load(os.path.join(golden_dir, "sam_reference.npz"))
    W = sam.init_weights(seed=int(g["weight_seed"]))
    img = torch.from_numpy(g["image"])
    with torch.no_grad():
        emb = sam.image_encoder(img, W)
    assert np.abs(emb.numpy() - g["image_embedding"]).max() < 1e-4
    pe = sam.dense_pe(W)
    assert np.array_equal(pe.numpy(), g["dense_pe"])
    text = torch.from_numpy(g["text_embeds"])
    emb2 = torch.from_numpy(np.concatenate([g["image_embedding"], g["image_embedding"][..., ::-1]], 0).copy())
    sp, de = sam.prompt_encoder_text(text, W)
    with torch.no_grad():
        masks, iou = sam.mask_decoder(emb2, pe, sp, de, W)
    assert np.abs(masks.numpy() - g["low_res_masks"]).max() < 1e-4
    assert np.abs(iou.numpy() - g["iou_pred"]).max() < 1e-5


def test_mask_head_oracle_matches_reference_golden(golden_dir):
    g = np.load(os.path.join(golden_dir, "mask_head_reference.npz"))
    for i in range(int(g["pp_count"])):
        x = torch.from_numpy(g[f"pp{i}_in"])
        out = ops.postprocess_masks(x, tuple(g[f"pp{i}_input_size"]), tuple(int(v) for v in g[f"pp{i}_original_size"]))
        assert np.array_equal(out.numpy(), g[f"pp{i}_out"]), i
    pred, gt, piou = (torch.from_numpy(g[k]) for k in ("loss_pred", "loss_gt", "loss_pred_iou"))
    for i in range(pred.shape[0]):
        gm = gt[i].unsqueeze(0)
        terms = [ops.sigmoid_ce_loss(pred[i], gm, 1).item(), ops.dice_loss(pred[i], gm).item(),
                 ops.mask_iou_loss(pred[i], gm, piou[i]).item(), ops.focal_loss(pred[i], gm).item()]
        assert np.allclose(terms, g["loss_terms"][i], rtol=1e-6, atol=1e-7)
    b, counts, iou, dice = ops.threshold_iou(pred[0, 0], gt[0])
    assert np.array_equal(b.numpy(), g["thr_mask"]) and list(counts) == list(g["thr_counts"])
    assert abs(iou - float(g["thr_iou"])) < 1e-12 and abs(dice - float(g["thr_dice"])) < 1e-12


def test_validate_metrics_match_reference_functions(golden_dir):
    """validate()'s per-sample numbers from the four threshold counts vs the reference's intersectionAndUnionGPU / calculate_iou
    run on the same masks (mask_head_reference.npz: validate_metrics); the product's host function must agree too."""
    from medplib_amd import metrics
    g = np.load(os.path.join(golden_dir, "mask_head_reference.npz"))
    pred, gt = torch.from_numpy(g["loss_pred"]), torch.from_numpy(g["loss_gt"])
    meters = metrics.SegMeters()
    for i in range(pred.shape[0]):
        _, counts, _, _ = ops.threshold_iou(pred[i, 0], gt[i])
        for fn in (ops.validate_metrics, metrics.metrics_from_counts):
            m = fn(counts, gt[i].numel())
            got = np.concatenate([m["intersection"], m["union"], m["acc_iou"], [m["iou"], m["dice"]]])
            assert np.allclose(got, g["validate_metrics"][i], rtol=1e-6, atol=0), (i, got, g["validate_metrics"][i])
        meters.update(m)
    s = meters.summary()
    vm = g["validate_metrics"]
    assert abs(s["giou"] - vm[:, 5].mean()) < 1e-6 and abs(s["dice"] - vm[:, 7].mean()) < 1e-6
    assert abs(s["ciou"] - vm[:, 1].sum() / (vm[:, 3].sum() + 1e-10)) < 1e-6


def _glue_case(g, tag):
    from oracle import llm
    W = {k[len(tag) + 3:]: torch.from_numpy(g[k]) for k in g.files if k.startswith(f"{tag}_W_")}
    ids, labels, att = (torch.from_numpy(g[f"{tag}_{k}"]) for k in ("ids", "labels", "att"))
    images = torch.from_numpy(g[f"{tag}_images"])
    P, Cv = 6, 4
    feats = torch.nn.functional.linear(images.flatten(1)[:, :P * Cv].reshape(images.shape[0], P, Cv),
                                       W["model.mm_projector.weight"], W["model.mm_projector.bias"])
    n_tok = P
    if "model.mm_token_compressor.proj.weight" in W:
        n_tok = 4
        feats = llm.token_compressor(feats, W, n_tok)
    per_token, types, lengths, mask_feats = f"{tag}_n_images" in g.files, None, None, None
    feat_list = list(feats) if per_token else feats
    if f"{tag}_types" in g.files:
        Wm = llm.init_icl_weights(feats.shape[-1], int(g[f"{tag}_icl_seed"]))
        mask_feats = llm.mask_token_encoder(torch.from_numpy(g[f"{tag}_mask_images"]), Wm, 3)
        types = [["mask" if t else "image" for t in row] for row in g[f"{tag}_types"]]
        lengths = [list(map(int, row)) for row in g[f"{tag}_lengths"]]
        feat_list = llm.combine_icl_features(list(feats), list(mask_feats), types)
    return dict(W=W, ids=ids, labels=labels, att=att, feats=feats, mask_feats=mask_feats, feat_list=feat_list, per_token=per_token,
                types=types, lengths=lengths, n_tok=n_tok)


def test_glue_oracle_matches_executed_reference(golden_dir):
    """Splice (3 layouts incl. ICL separate mode with the mask encoder), <SEG> mask, TokenCompressor, MaskTokenEncoder vs the
    outputs of the reference's own functions (oracle/make_golden.py: golden_glue)."""
    from oracle import llm
    g = np.load(os.path.join(golden_dir, "glue_reference.npz"))
    for tag in ("A", "B", "C"):
        c = _glue_case(g, tag)
        with torch.no_grad():
            att, emb, lab = llm.prepare_inputs_labels_for_multimodal(c["ids"], c["att"], c["labels"], c["feat_list"],
                                                                     c["W"]["model.embed_tokens.weight"], c["per_token"])
            seg = llm.build_seg_token_mask(c["ids"], 33, c["n_tok"], c["lengths"])
        assert np.array_equal(lab.numpy(), g[f"{tag}_new_labels"]) and np.array_equal(att.numpy(), g[f"{tag}_new_att"]), tag
        assert np.array_equal(seg.numpy(), g[f"{tag}_seg_mask"]), tag
        assert np.abs(emb.numpy() - g[f"{tag}_embeds"]).max() < 1e-6, tag
    W = llm.init_icl_weights(int(g["icl_hidden"]), int(g["icl_weight_seed"]))
    assert abs(sum(float(v.double().sum()) for v in W.values()) - float(g["icl_weight_checksum"])) < 1e-6, "seeded weights drifted"
    with torch.no_grad():
        y = llm.token_compressor(torch.from_numpy(g["tc_x"]), W, 256)
        mk = torch.from_numpy(np.unpackbits(g["me_mask_bits"])[: 2 * 336 * 336].reshape(2, 1, 336, 336).astype(np.float32))
        ym = llm.mask_token_encoder(mk, W, 64)
    assert np.abs(y.numpy() - g["tc_y"]).max() < 1e-6
    assert np.abs(ym.numpy() - g["me_y"]).max() < 1e-5


def _lisa_golden(golden_dir):
    from oracle import make_golden as MG, model as OM
    g = np.load(os.path.join(golden_dir, "lisa_forward_reference.npz"))
    cfg = MG.lisa_tiny_cfg()
    W = OM.init_hf_weights(cfg, seed=int(g["weight_seed"]))
    assert abs(sum(float(v.double().sum()) for v in W.values()) - float(g["weight_checksum"])) < 1e-6, "seeded weights drifted"
    return g, cfg, W, MG.lisa_cases(cfg)


def test_model_forward_oracle_matches_executed_reference_lisa(golden_dir):
    """The END-TO-END pin: the oracle's `model_forward` against what the reference's own
    `LISAForCausalLM(config).train().model_forward(...)` (model/LISA.py:260-471 over medplib_llama.py:55-148, medplib_arch.py:217-527,
    the SAM-Med2D modules and HF Llama / CLIP) returned on the same seeded weights and batches — oracle/make_golden.py: golden_lisa.
    Three batches: standard, ragged right padding, and valid_mask_bool = [[True], [True, True], []] (expand_embedding,
    MedPLIB.py:292-308 = LISA.py:228-239).  Losses 1e-5, last hidden state 2e-5, gradient (sum, norm) of EVERY tensor
    loss.backward() reaches 2e-3 relative, stored full gradients 2e-3 of the tensor's largest entry."""
    from oracle import model as OM
    g, cfg, W, cases = _lisa_golden(golden_dir)
    stat_keys = [str(k) for k in g["grad_stat_keys"]]
    for name, b in cases.items():
        chk = float(b["images"].double().sum()) + float(b["images_clip"].double().sum()) + float(b["input_ids"].sum())
        assert abs(chk - float(g[f"{name}_input_checksum"])) < 1e-6, "seeded batch drifted"
        Wr = {k: (v.clone().requires_grad_() if k in stat_keys else v) for k, v in W.items()}
        out, inter = OM.model_forward(b, Wr, cfg, training=True, llm_grad=True, return_intermediates=True)
        out["loss"].backward()
        got = np.array([float(out[k].detach()) for k in ops.LOSS_KEYS])
        assert np.abs(got - g[f"{name}_losses"]).max() < 1e-5, (name, got, g[f"{name}_losses"])
        assert np.abs(inter["hidden"][:, -72:].detach().numpy() - g[f"{name}_hidden_tail"]).max() < 2e-5, name
        stats = np.array([[float(Wr[k].grad.double().sum()), float(Wr[k].grad.double().norm())] for k in stat_keys])
        ref = g[f"{name}_grad_stats"]
        scale = ref[:, 1:2] + 1e-4      # the gradient norm scales both statistics (floor: k_proj.bias gradients are 0 + noise)
        assert (np.abs(stats - ref) / scale).max() < 2e-3, (name, stat_keys[int((np.abs(stats - ref) / scale).max(1).argmax())])
        for k in g.files:
            if k.startswith(f"{name}_grad_") and k != f"{name}_grad_stats":
                pk = k[len(name) + 6:]
                r = g[k]
                assert np.abs(Wr[pk].grad.numpy() - r).max() <= 2e-3 * np.abs(r).max() + 1e-9, (name, pk)
        pm = np.concatenate([p.detach().reshape(-1).numpy() for p in inter["pred_masks"]])
        assert np.abs(pm - g[f"{name}_pred_masks"].astype(np.float32)).max() < 2e-3 * np.abs(pm).max() + 1e-3, name
        if name == "multimask":
            assert len(inter["pred_masks"]) == 3 and [tuple(p.shape[-2:]) for p in inter["pred_masks"]] == [(96, 80), (64, 72), (96, 80)]


def test_oracle_decoder_layer_at_true_dims_matches_hf_golden(golden_dir):
    """One dense decoder layer + final norm at the 7B dims vs 64 rows the installed HuggingFace LlamaModel produced on the same
    seeded weights (oracle/make_golden.py: golden_llama_layer; transformers 5.15 — the reference pins 4.31)."""
    from medplib_amd.model.config import MedPLIBConfig
    from oracle import llm, model as OM
    g = np.load(os.path.join(golden_dir, "llama_layer_truedims.npz"))
    cfg = MedPLIBConfig.medplib_7b(num_hidden_layers=1, vocab_size=1024, moe_enable=False, moe_gate_sampling=False)
    W, gen = OM.init_decoder_layer_weights(cfg, seed=int(g["weight_seed"]))
    emb, kv = OM.decoder_layer_inputs(cfg, gen)
    with torch.no_grad():
        out, _ = llm.llama_forward(emb.float(), kv, W, cfg, training=True)
    got = out.view(-1, cfg.hidden_size)[torch.from_numpy(g["rows"])].numpy()
    assert np.abs(got - g["hidden_rows"]).max() < 5e-4


def test_evaluate_oracle_matches_executed_reference(golden_dir):
    """`oracle.model.evaluate` vs the EXECUTED `LISAForCausalLM.evaluate` (model/LISA.py:473-555, the dense twin of MedPLIB.py:574-680;
    tests/golden/lisa_evaluate_reference.npz, oracle/make_golden.py: golden_evaluate): greedy token ids bit-equal over 16-20 new
    tokens (3 when the model emits EOS), the <SEG> pick rules (none generated -> position -2, two in the prompt -> the first, <SEG>
    generated by the model itself), the mask within 1e-4."""
    from oracle import make_golden as MG
    from oracle import model as OM
    g = np.load(os.path.join(golden_dir, "lisa_evaluate_reference.npz"))
    cfg = MG.lisa_tiny_cfg()
    W = OM.init_hf_weights(cfg, seed=int(g["weight_seed"]))
    assert [str(c) for c in g["cases"]] == list(MG.EVAL_CASES)
    for name in MG.EVAL_CASES:
        edit = g[f"{name}_edit"]
        b, Wc, n_new, _ = MG.evaluate_case(cfg, W, name, None if edit[0] < 0 else edit)
        assert abs(float(b["images"].double().sum()) + float(b["images_clip"].double().sum()) + float(b["input_ids"].sum())
                   - float(g[f"{name}_input_checksum"])) < 1e-6
        ids, masks = OM.evaluate(b, Wc, cfg, max_new_tokens=n_new)
        assert np.array_equal(ids.numpy(), g[f"{name}_output_ids"]), name
        assert np.abs(masks[0].numpy() - g[f"{name}_pred_mask"]).max() < 1e-4, name
    n_in = 40
    assert g["fallback_output_ids"].shape[1] - n_in == 20 and cfg.seg_token_idx not in g["fallback_output_ids"][0]
    assert cfg.seg_token_idx in g["seg_generated_output_ids"][0, n_in:] and g["eos_output_ids"][0, -1] == 2
