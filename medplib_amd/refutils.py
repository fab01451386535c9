"""The helper names the reference's drivers import from `utils.utils` (train_ds_medplib.py:24-26, model/eval/vqa_infer.py:27-29):
token strings, `AverageMeter` / `ProgressMeter` / `Summary`, `dict_to_cuda`, `intersectionAndUnionGPU`, `ADD_OTHERS_TOKENS`.
The repo-root `utils/utils.py` re-exports this module, so `from utils.utils import ...` keeps working in an unchanged driver.

Same names, same arguments, same attributes and printed formats (reference utils/utils.py:7-139); the bodies are this build's:
class counts by `bincount` instead of three `histc` passes, one packed reduction per meter, index tensors left on the host by
`dict_to_cuda` (the splice plan is host work in this build — the model accepts both)."""
import enum

import numpy as np
import torch
import torch.distributed as dist

IGNORE_INDEX = -100
IMAGE_TOKEN_INDEX = -200
REGION_TOKEN_INDEX = -300
DEFAULT_IMAGE_TOKEN = "<image>"
DEFAULT_IMAGE_PATCH_TOKEN = "<im_patch>"
DEFAULT_IM_START_TOKEN = "<im_start>"
DEFAULT_IM_END_TOKEN = "<im_end>"
DEFAULT_REGION_REFER_TOKEN_0 = "<region>"
DEFAULT_REGION_REFER_TOKEN_1 = "</region>"
ADD_OTHERS_TOKENS = ["<SEG>", "<ref>", "</ref>", "<region>", "</region>", "<sr>", "</sr>", "<mask>", "</mask>"]

HOST_SIDE_KEYS = ("input_ids", "labels", "attention_mask", "offset")


class Summary(enum.Enum):
    NONE = 0
    AVERAGE = 1
    SUM = 2
    COUNT = 3


_SUMMARY_FIELD = {Summary.AVERAGE: "avg", Summary.SUM: "sum", Summary.COUNT: "count"}


class AverageMeter:
    """Running value / sum / count / average of a scalar or of a small numpy vector (utils/utils.py:28-89)."""

    def __init__(self, name, fmt=":f", summary_type=Summary.AVERAGE):
        self.name, self.fmt, self.summary_type = name, fmt, summary_type
        self.reset()

    def reset(self):
        self.val = self.avg = self.sum = self.count = 0

    def update(self, val, n=1):
        self.val = val
        self.sum = self.sum + val * n
        self.count += n
        self.avg = self.sum / self.count

    def all_reduce(self):
        """SUM over the ranks of (sum, count); afterwards avg = sum / (count + 1e-5) as in the reference (:49-70).  A no-op without
        a process group (the reference would raise: its drivers always run under the launcher)."""
        vec = isinstance(self.sum, np.ndarray)
        packed = (list(np.asarray(self.sum, dtype=np.float64).ravel()) if vec else [float(self.sum)]) + [float(self.count)]
        if dist.is_available() and dist.is_initialized():
            dev = "cuda" if (torch.cuda.is_available() and dist.get_backend() == "nccl") else "cpu"
            t = torch.tensor(packed, dtype=torch.float32, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
            packed = t.cpu().tolist()
        self.count = packed[-1]
        self.sum = np.asarray(packed[:-1], dtype=np.float32) if vec else packed[0]
        self.avg = self.sum / (self.count + 1e-5)

    def __str__(self):
        return ("{name} {val" + self.fmt + "} ({avg" + self.fmt + "})").format(**vars(self))

    def summary(self):
        if self.summary_type is Summary.NONE:
            return ""
        if self.summary_type not in _SUMMARY_FIELD:
            raise ValueError("invalid summary type %r" % self.summary_type)
        return ("{name} {" + _SUMMARY_FIELD[self.summary_type] + ":.3f}").format(**vars(self))


class ProgressMeter:
    """`[ 12/100]`-prefixed line of meters (utils/utils.py:107-126)."""

    def __init__(self, num_batches, meters, prefix=""):
        width = len(str(num_batches // 1))
        self.batch_fmtstr = "[{:" + str(width) + "d}/" + ("{:" + str(width) + "d}").format(num_batches) + "]"
        self.meters, self.prefix = meters, prefix

    def display(self, batch):
        print("\t".join([self.prefix + self.batch_fmtstr.format(batch)] + [str(m) for m in self.meters]))

    def display_summary(self):
        print(" ".join([" *"] + [m.summary() for m in self.meters]))


def intersectionAndUnionGPU(output, target, K, ignore_index=255):
    """Per-class |pred & gt|, |pred | gt|, |gt| for label maps with values in [0, K) (utils/utils.py:92-104) -> three float tensors
    [K].  Like the reference, pixels whose TARGET is `ignore_index` are written as `ignore_index` into `output` IN PLACE (its callers
    pass a clone) and count for no class."""
    assert output.dim() in [1, 2, 3]
    assert output.shape == target.shape
    out, tgt = output.view(-1), target.view(-1)
    out[tgt == ignore_index] = ignore_index

    def per_class(x):
        x = x[(x >= 0) & (x < K)].long()
        return torch.bincount(x, minlength=K)[:K].to(torch.float32)
    area_intersection = per_class(out[out == tgt])
    area_output, area_target = per_class(out), per_class(tgt)
    return area_intersection, area_output + area_target - area_intersection, area_target


def dict_to_cuda(input_dict, device=None):
    """Move a collated batch to the current GPU in place and return it (utils/utils.py:129-139): tensors and non-empty lists of
    tensors.  The index tensors of HOST_SIDE_KEYS stay where they are: this build plans the splice on the host and the model
    accepts them on either side (the reference's one-argument call form is unchanged)."""
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    for key, value in input_dict.items():
        if key in HOST_SIDE_KEYS:
            continue
        if torch.is_tensor(value):
            input_dict[key] = value.to(dev, non_blocking=True)
        elif isinstance(value, list) and value and torch.is_tensor(value[0]):
            input_dict[key] = [t.to(dev, non_blocking=True) for t in value]
    return input_dict
