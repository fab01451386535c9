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
