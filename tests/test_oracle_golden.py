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
