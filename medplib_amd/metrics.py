"""Segmentation metrics of the reference's `validate()` loop (train_ds_medplib.py:721-800) / `validate_seg` (vqa_infer.py:565-588)
from the integer counts the threshold kernel returns (mp_mask_threshold_iou: |pred|, |gt|, |pred & gt|, |pred | gt| with
pred = sigmoid(logit) > 0.1).  The per-pixel work stays on the GPU; what comes to the host is four integers per mask."""
import numpy as np
import torch

from . import ops


def metrics_from_counts(counts, n_pixels):
    """One sample: intersectionAndUnionGPU(K = 2) per-class intersection / union (utils/utils.py:92-104), acc_iou with the
    "no-object target" rule (train_ds_medplib.py:766-767), IoU (calculate_iou :702-719) and Dice = 2 IoU / (1 + IoU) (:771-772)."""
    _, _, inter, union = (int(c) for c in counts)
    I = np.array([n_pixels - union, inter], dtype=np.float32)
    U = np.array([n_pixels - inter, union], dtype=np.float32)
    acc = I / (U + np.float32(1e-5))
    acc[U == 0] += 1.0
    iou = 0.0 if union == 0 else float(np.float32(inter) / np.float32(union))
    return {"intersection": I, "union": U, "acc_iou": acc, "iou": iou, "dice": 2 * iou / (1 + iou)}


class SegMeters:
    """The five meters of validate(): running sums of intersection / union / acc_iou (per class) and of IoU / Dice, reduced over
    ranks with one SUM all-reduce (`AverageMeter.all_reduce`, utils/utils.py:49-70)."""

    def __init__(self):
        self.intersection = np.zeros(2, np.float64); self.union = np.zeros(2, np.float64); self.acc_iou = np.zeros(2, np.float64)
        self.iou = 0.0; self.dice = 0.0; self.count = 0

    def update(self, m):
        self.intersection += m["intersection"]; self.union += m["union"]; self.acc_iou += m["acc_iou"]
        self.iou += m["iou"]; self.dice += m["dice"]; self.count += 1

    def all_reduce(self, device="cpu"):
        import torch.distributed as dist
        if not dist.is_initialized():
            return
        t = torch.tensor(list(self.intersection) + list(self.union) + list(self.acc_iou) + [self.iou, self.dice, self.count],
                         dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        v = t.cpu().numpy()
        self.intersection, self.union, self.acc_iou = v[0:2], v[2:4], v[4:6]
        self.iou, self.dice, self.count = float(v[6]), float(v[7]), int(v[8])

    def summary(self):
        """(gIoU, cIoU, mean IoU, mean Dice) as validate() reports them: cIoU = (sum I / (sum U + 1e-10))[1], gIoU = mean acc_iou[1]."""
        n = max(self.count, 1)
        ciou = (self.intersection / (self.union + 1e-10))[1]
        return {"giou": float(self.acc_iou[1] / n), "ciou": float(ciou), "iou": self.iou / n, "dice": self.dice / n}


@torch.no_grad()
def validate_batch(model, batch, meters, threshold=0.1):
    """One validation sample: forward in inference mode ({pred_masks, gt_masks}, MedPLIB.py:507-511), threshold + counts on the GPU,
    meters on the host — the body of validate()'s loop (train_ds_medplib.py:745-772).  Returns the per-sample metric dict."""
    out = model(**dict(batch, inference=True))
    pred, gt = out["pred_masks"][0], out["gt_masks"][0]
    n = pred.shape[0]
    gt = gt.reshape(n, -1).to(device=pred.device, dtype=torch.float32).contiguous()
    _, counts = ops.mask_threshold_iou(pred.reshape(n, -1).contiguous(), gt, threshold)
    m = None
    for c in counts.cpu().tolist():
        m = metrics_from_counts(c, pred[0].numel())
        meters.update(m)
    return m
