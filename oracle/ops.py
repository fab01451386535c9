"""ORACLE — TEST INFRASTRUCTURE ONLY.  Not part of the product path.

CPU fp32 (torch) restatement of the per-kernel arithmetic on the MedPLIB hot path.  Only `tests/`,
`__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import this package; `medplib_amd/` never does.

Every function cites the reference site (paths relative to the reference checkout) whose arithmetic it restates.
Third-party arithmetic (transformers==4.31.0 Llama/CLIP, deepspeed==0.13.1 MoE) is restated from the published
algorithms; where no reference test pins those, the docstring says "parity unpinned"."""
import math

import torch
import torch.nn.functional as F

IGNORE_INDEX = -100          # utils/utils.py:7
IMAGE_TOKEN_INDEX = -200     # utils/utils.py:8
REGION_TOKEN_INDEX = -300    # utils/utils.py:9


# ----------------------------------------------------------------------------------------------- trunk primitives
def linear(x, w, b=None):
    """nn.Linear: x @ w^T + b (weight [out, in])."""
    return F.linear(x, w, b)


def rmsnorm(x, w, eps):
    """HF-4.31 LlamaRMSNorm (transformers pinned at requirements.txt:137; call sites medplib_moe_llama.py:121,138,286).
    parity unpinned (third-party); cross-checked against the installed transformers' LlamaRMSNorm in tests."""
    v = x.float().pow(2).mean(-1, keepdim=True)
    return w * (x.float() * torch.rsqrt(v + eps))


def rope_tables(seq, head_dim, theta=10000.0):
    """HF LlamaRotaryEmbedding: inv_freq = theta^(-2i/d); emb = cat(freqs, freqs)  (SURVEY Appendix A.1)."""
    inv = 1.0 / (theta ** (torch.arange(0, head_dim, 2, dtype=torch.float32) / head_dim))
    t = torch.arange(seq, dtype=torch.float32)
    freqs = torch.outer(t, inv)          # [seq, d/2]
    return freqs.cos(), freqs.sin()


def rope(x, cos, sin):
    """x [B,S,H,D]; half-split rotate_half = cat(-x2, x1) (HF apply_rotary_pos_emb)."""
    d = x.shape[-1] // 2
    c = torch.cat([cos, cos], -1)[None, : x.shape[1], None, :]
    s = torch.cat([sin, sin], -1)[None, : x.shape[1], None, :]
    rot = torch.cat([-x[..., d:], x[..., :d]], -1)
    return x * c + rot * s


def swiglu(gate, up):
    """HF LlamaMLP: silu(gate) * up."""
    return F.silu(gate) * up


def attention(q, k, v, causal=False, key_valid=None, bias=None, scale=None):
    """Eager softmax attention, q/k/v [B,S,H,D] -> [B,Sq,H*D].
    HF-4.31 LlamaAttention eager semantics: masked scores receive zero weight (additive finfo.min),
    softmax in fp32, padded query rows still produce values.  bias [B,H,Sq,Sk] is added unscaled (SAM rel-pos,
    image_encoder.py:290-293)."""
    B, Sq, H, D = q.shape
    Sk = k.shape[1]
    scale = D ** -0.5 if scale is None else scale
    s = torch.einsum("bqhd,bkhd->bhqk", q.float(), k.float()) * scale
    if bias is not None:
        s = s + bias
    neg = torch.finfo(torch.float32).min
    if causal:
        cm = torch.ones(Sq, Sk, dtype=torch.bool).tril()
        s = s.masked_fill(~cm[None, None], neg)
    if key_valid is not None:
        s = s.masked_fill(~key_valid.bool()[:, None, None, :], neg)
    p = torch.softmax(s, -1)
    return torch.einsum("bhqk,bkhd->bqhd", p, v.float()).reshape(B, Sq, H * D)


def quick_gelu(x):
    """HF CLIP `quick_gelu`: x * sigmoid(1.702 x)."""
    return x * torch.sigmoid(1.702 * x)


# ----------------------------------------------------------------------------------------------- SAM rel-pos helpers
def get_rel_pos(q_size, k_size, rel_pos):
    """model/segment_anything_med2d/modeling/image_encoder.py:348-378 (no interpolation needed at 256 px)."""
    max_rel_dist = int(2 * max(q_size, k_size) - 1)
    if rel_pos.shape[0] != max_rel_dist:
        r = F.interpolate(rel_pos.reshape(1, rel_pos.shape[0], -1).permute(0, 2, 1), size=max_rel_dist, mode="linear")
        rel_pos = r.reshape(-1, max_rel_dist).permute(1, 0)
    q_coords = torch.arange(q_size)[:, None] * max(k_size / q_size, 1.0)
    k_coords = torch.arange(k_size)[None, :] * max(q_size / k_size, 1.0)
    rel = (q_coords - k_coords) + (k_size - 1) * max(q_size / k_size, 1.0)
    return rel_pos[rel.long()]


def decomposed_rel_pos(q, rel_pos_h, rel_pos_w, hw):
    """image_encoder.py:381-421: q [B', h*w, C] (unscaled) -> rel_h [B', h*w, h], rel_w [B', h*w, w]."""
    h, w = hw
    Rh = get_rel_pos(h, h, rel_pos_h)
    Rw = get_rel_pos(w, w, rel_pos_w)
    B = q.shape[0]
    r_q = q.reshape(B, h, w, -1)
    rel_h = torch.einsum("bhwc,hkc->bhwk", r_q, Rh)
    rel_w = torch.einsum("bhwc,wkc->bhwk", r_q, Rw)
    return rel_h.reshape(B, h * w, h), rel_w.reshape(B, h * w, w)


# ----------------------------------------------------------------------------------------------- mask head
def postprocess_masks(masks, input_size, original_size):
    """model/MedPLIB.py:682-701 — Python-slice 'crop' (clamped / wrapping exactly as Python does) then bilinear resize."""
    if masks.dim() == 3:
        masks = masks.unsqueeze(0)
    pad_h = masks.shape[-2] - input_size[0]
    pad_w = masks.shape[-1] - input_size[1]
    pad_top, pad_left = pad_h // 2, pad_w // 2
    oh, ow = masks.shape[-2] - pad_h, masks.shape[-1] - pad_w
    masks = masks[:, :, pad_top:pad_top + oh, pad_left:pad_left + ow]
    return F.interpolate(masks, original_size, mode="bilinear", align_corners=False)


def sigmoid_ce_loss(inputs, targets, num_masks):
    """model/MedPLIB.py:107-124."""
    loss = F.binary_cross_entropy_with_logits(inputs, targets, reduction="none")
    return loss.flatten(1, 2).mean(1).sum() / (num_masks + 1e-8)


def dice_loss(inputs, targets, eps=1e-6):
    """model/MedPLIB.py:71-104."""
    inputs = torch.sigmoid(inputs)
    inputs = inputs.view(inputs.size(0), -1)
    targets = targets.view(targets.size(0), -1)
    inter = (inputs * targets).sum(-1)
    union = inputs.sum(-1) + targets.sum(-1)
    return (1 - (2.0 * inter + eps) / (union + eps)).mean()


def mask_iou_loss(pred_mask, gt, pred_iou):
    """model/MedPLIB.py:26-44 (MaskIoULoss)."""
    p = torch.sigmoid(pred_mask)
    inter = torch.sum(p * gt)
    union = torch.sum(p) + torch.sum(gt) - inter
    iou = (inter + 1e-7) / (union + 1e-7)
    return torch.mean((iou - pred_iou) ** 2)


def focal_loss(pred, mask, gamma=2.0, alpha=0.25):
    """model/MedPLIB.py:46-69 (FocalLoss)."""
    p = torch.sigmoid(pred)
    num_pos = torch.sum(mask)
    num_neg = mask.numel() - num_pos
    w_pos = (1 - p) ** gamma
    w_neg = p ** gamma
    loss_pos = -alpha * mask * w_pos * torch.log(p + 1e-12)
    loss_neg = -(1 - alpha) * (1 - mask) * w_neg * torch.log(1 - p + 1e-12)
    return (torch.sum(loss_pos) + torch.sum(loss_neg)) / (num_pos + num_neg + 1e-12)


def combine_mask_losses(pred_masks, gt_masks, pred_ious, ce_loss, weights):
    """model/MedPLIB.py:515-572: per-mask losses, /(num_masks+1e-8), weights, dict of 10 scalars.
    pred_masks: list of [1,H,W]; gt_masks: list of [H,W]; pred_ious: list of [1]; weights = dict ce/bce/dice/iou/focal."""
    ce = ce_loss * weights["ce"]
    bce = dice = iou = focal = 0
    num = 0
    for pm, gm, pi in zip(pred_masks, gt_masks, pred_ious):
        g = gm.unsqueeze(0)
        bce = bce + sigmoid_ce_loss(pm, g, num_masks=g.shape[0]) * g.shape[0]
        dice = dice + dice_loss(pm, g) * g.shape[0]
        iou = iou + mask_iou_loss(pm, g, pi) * g.shape[0]
        focal = focal + focal_loss(pm, g) * g.shape[0]
        num += g.shape[0]
    u_bce, u_dice = bce / (num + 1e-8), dice / (num + 1e-8)
    u_iou, u_focal = iou / (num + 1e-8), focal / (num + 1e-8)
    m_bce, m_dice = weights["bce"] * u_bce, weights["dice"] * u_dice
    m_iou, m_focal = weights["iou"] * u_iou, weights["focal"] * u_focal
    mask_loss = m_bce + m_dice + m_iou + m_focal
    return {
        "loss": ce + mask_loss, "ce_loss": ce, "mask_bce_loss": m_bce, "mask_dice_loss": m_dice, "mask_loss": mask_loss,
        "unscale_mask_bce_loss": u_bce, "unscale_mask_dice_loss": u_dice,
        "unscale_mask_loss": u_bce + u_dice + u_iou + u_focal,
        "unscale_mask_iou_loss": u_iou, "unscale_mask_focal_loss": u_focal,
    }


LOSS_KEYS = ["loss", "ce_loss", "mask_bce_loss", "mask_dice_loss", "mask_loss", "unscale_mask_bce_loss",
             "unscale_mask_dice_loss", "unscale_mask_loss", "unscale_mask_iou_loss", "unscale_mask_focal_loss"]


def threshold_iou(pred_logits, gt, thr=0.1):
    """train_ds_medplib.py:702-719,750,771-772 / vqa_infer.py:565-588:
    bin = sigmoid(x) > 0.1; IoU = |and|/|or| (0 if union empty); Dice = 2 IoU / (1 + IoU)."""
    b = torch.sigmoid(pred_logits) > thr
    g = gt.bool()
    inter = torch.logical_and(b, g).sum().item()
    union = torch.logical_or(b, g).sum().item()
    iou = 0.0 if union == 0 else inter / union
    return b, (int(b.sum()), int(g.sum()), inter, union), iou, 2 * iou / (1 + iou)


def mask_cut_report(pred, ref, gt, thr=0.1):
    """How two mask-logit maps compare where a comparison can fail: at the reference's threshold `sigmoid(x) > 0.1`
    (train_ds_medplib.py:750; logit cut log(0.1 / 0.9) = -2.197) and at logit 0.  Per cut: the positive-pixel fraction on both
    sides, the number of pixels whose thresholded value differs, the number of reference pixels within the measured max |dlogit| of
    the cut (the only pixels that MAY differ: `flipped <= near_cut` is the bit-exactness statement for mask indices under a
    stated logit tolerance), and the Dice against the ground truth on both sides.  pred / ref / gt: [H, W] (or [1, H, W])."""
    import math
    pred, ref, g = pred.float().reshape(-1), ref.float().reshape(-1), gt.reshape(-1).bool()
    err = (pred - ref).abs()
    out = {"max_abs_dlogit": float(err.max()), "mean_abs_dlogit": float(err.mean()), "pixels": int(ref.numel())}
    for name, cut in (("cut_ref", math.log(thr / (1.0 - thr))), ("cut_zero", 0.0)):
        bp, br = pred > cut, ref > cut
        def dice(b):
            inter, union = int((b & g).sum()), int((b | g).sum())
            iou = 0.0 if union == 0 else inter / union
            return 2 * iou / (1 + iou)
        out[name] = {"logit_cut": cut, "pos_frac_pred": float(bp.float().mean()), "pos_frac_ref": float(br.float().mean()),
                     "flipped": int((bp != br).sum()), "near_cut": int(((ref - cut).abs() <= out["max_abs_dlogit"]).sum()),
                     "dice_pred": dice(bp), "dice_ref": dice(br)}
        out[name]["abs_ddice"] = abs(out[name]["dice_pred"] - out[name]["dice_ref"])
    return out


def validate_metrics(counts, n_pixels):
    """Per-sample metrics of validate() (train_ds_medplib.py:745-772) from the four integer counts of threshold_iou
    (|pred|, |gt|, |pred & gt|, |pred | gt|) for a binary target without ignore pixels: intersectionAndUnionGPU (utils/utils.py:
    92-104) with K = 2 gives intersection = [N - |or|, |and|], union = [N - |and|, |or|]; acc_iou = I / (U + 1e-5), +1 where the
    union is empty; IoU = |and| / |or| (0 when empty, calculate_iou :702-719); Dice = 2 IoU / (1 + IoU)."""
    import numpy as np
    _, _, inter, union = (int(c) for c in counts)
    I = np.array([n_pixels - union, inter], dtype=np.float32)
    U = np.array([n_pixels - inter, union], dtype=np.float32)
    acc = I / (U + np.float32(1e-5))
    acc[U == 0] += 1.0
    iou = 0.0 if union == 0 else float(np.float32(inter) / np.float32(union))
    return {"intersection": I, "union": U, "acc_iou": acc, "iou": iou, "dice": 2 * iou / (1 + iou)}


def cross_entropy_filtered(logits, labels):
    """medplib_moe_llama.py:392-408: shift, drop batch rows whose shifted labels are all -100, mean CE over the rest.
    logits [B,S,V] fp32, labels [B,S]."""
    sl = logits[..., :-1, :]
    lab = labels[..., 1:]
    keep = (lab != IGNORE_INDEX).any(dim=1)
    sl, lab = sl[keep], lab[keep]
    return F.cross_entropy(sl.reshape(-1, sl.shape[-1]), lab.reshape(-1), ignore_index=IGNORE_INDEX)
