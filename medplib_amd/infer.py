"""Inference / evaluation entry point with the control flow of the reference's `model/eval/vqa_infer.py`: `validate_seg` (:488-633 —
per sample: prompt cut after the last "ASSISTANT:" colon, `model.evaluate(...)` = greedy decode + one mask, threshold 0.1, IoU /
Dice meters, per-modality breakdown) and the VQA loop (:394-486 — prompt cut the same way, `model.generate`, one JSON line per
sample in `--answers_file`).  `--dataset medplib_amd.dataset:val_from_args` reads the reference's JSON files (`--val_data_path
--image_folder --tokenizer_path`); `synthetic` runs the seeded generator the tests and benchmarks use.  String metrics
(BLEU / F1 over the answers file) are the reference's own scripts."""
import argparse
import importlib
import json
import os

import torch

from . import metrics, ops
from .train import SyntheticDataset, build_model, dict_to_device

COLON_ID = 29901          # the ':' that ends "ASSISTANT:" in the llava_v1 prompt (vqa_infer.py:521-523)


def parse_args(argv=None):
    p = argparse.ArgumentParser(description="MedPLIB segmentation / VQA inference on MI355X")
    p.add_argument("--version", default="")
    p.add_argument("--vision_pretrained", default="")
    p.add_argument("--model_size", default="7b", choices=["7b", "tiny"])
    p.add_argument("--dataset", default="synthetic")
    p.add_argument("--eval_seg", action="store_true", default=True)
    p.add_argument("--eval_vqa", action="store_true", default=False)
    p.add_argument("--max_new_tokens", default=64, type=int)
    p.add_argument("--n_samples", default=4, type=int)
    p.add_argument("--lisa", action="store_true")
    p.add_argument("--seed", default=42, type=int)
    p.add_argument("--val_data_path", default="")
    p.add_argument("--image_folder", default="")
    p.add_argument("--tokenizer_path", default="")
    p.add_argument("--model_max_length", default=512, type=int)
    p.add_argument("--icl_enable", action="store_true", default=False)
    p.add_argument("--icl_mask_mode", default="overlay", choices=["overlay", "separate"])
    p.add_argument("--icl_mask_encoder", action="store_true", default=False)
    p.add_argument("--answers_file", default="")
    p.add_argument("--colon_token_id", default=COLON_ID, type=int)
    # the model flags build_model reads (train.py)
    for name, typ, dv in (("moe_enable", lambda s: s.lower() in ("1", "true"), True), ("num_experts", int, 2), ("top_k_experts", int, 1),
                          ("capacity_factor", float, 1.5), ("eval_capacity_factor", float, 2.0), ("min_capacity", int, 0),
                          ("router_aux_loss_coef", float, 0.0), ("ce_loss_weight", float, 1.0), ("dice_loss_weight", float, 0.5),
                          ("bce_loss_weight", float, 2.0), ("iou_loss_weight", float, 2.0), ("focal_loss_weight", float, 2.0), ("ep_size", int, 1), ("use_residual", bool, False)):
        p.add_argument("--" + name, type=typ, default=dv)
    p.add_argument("--train_mask_decoder", action="store_true", default=True)
    from .train import add_layout_flags
    add_layout_flags(p)
    return p.parse_args(argv)


def prompt_cut(ids, colon_id=COLON_ID):
    """Length of the prompt = up to and including the last ':' (the one closing "ASSISTANT:"; vqa_infer.py:426-428, 521-523)."""
    colon = (torch.as_tensor(ids) == colon_id).nonzero(as_tuple=True)
    return int(colon[1][-1]) + 1 if colon[1].numel() else ids.shape[1]


@torch.no_grad()
def validate_seg(val, model, device, max_new_tokens=64, colon_id=COLON_ID):
    """-> (mIoU, mDice, per-modality dict) over `val` (items = collated single-sample batches)."""
    model.eval()
    ious, dices, by_mod = [], [], {}
    for i in range(len(val)):
        b = dict_to_device(val[i], device)
        ids, att = b["input_ids"], b["attention_mask"]
        cut = prompt_cut(ids, colon_id)
        output_ids, pred_masks = model.evaluate(b["images_clip"], b["images"], ids[:, :cut], b["resize_list"], b["label_list"],
                                                max_new_tokens=max_new_tokens, attention_mask=att[:, :cut],
                                                mask_images=b.get("mask_images"), image_token_types=b.get("image_token_types"),
                                                image_token_lengths=b.get("image_token_lengths"),
                                                region_masks=b.get("region_masks") or (),
                                                valid_region_masks_bool=b.get("valid_region_masks_bool") or ())
        iou = 0.0
        if len(pred_masks) > 0:
            gt = b["masks_list"][0].reshape(1, -1).to(device=device, dtype=torch.float32).contiguous()
            _, counts = ops.mask_threshold_iou(pred_masks[0].reshape(1, -1).contiguous(), gt, 0.1)
            iou = metrics.metrics_from_counts(counts[0].cpu().tolist(), gt.numel())["iou"]
        dice = 2 * iou / (1 + iou)
        ious.append(iou); dices.append(dice)
        path = (b.get("image_paths") or [None])[0]
        mod = os.path.basename(path).split("_")[0] if path else "synthetic"
        by_mod.setdefault(mod, {"iou": [], "dice": []})
        by_mod[mod]["iou"].append(iou); by_mod[mod]["dice"].append(dice)
    miou, mdice = sum(ious) / max(len(ious), 1), sum(dices) / max(len(dices), 1)
    print("miou: {:.6f}, mDice: {:.6f}".format(miou, mdice))
    res = {m: {k: round(sum(v) / len(v), 6) for k, v in d.items()} for m, d in by_mod.items()}
    print(res)
    return miou, mdice, res


@torch.no_grad()
def run_vqa(val, model, device, max_new_tokens=64, colon_id=COLON_ID, answers_file="", tokenizer=None):
    """-> list of generated id tensors (prompt + continuation, as HF `generate` returns them).  With `answers_file`: one JSON line
    per sample {question_id, image_path, prompt, text | output_ids, gt} (vqa_infer.py:470-480); `text` needs a tokenizer with
    `batch_decode`, otherwise the new ids are written."""
    model.eval()
    outs = []
    fh = None
    if answers_file:
        os.makedirs(os.path.dirname(os.path.abspath(answers_file)), exist_ok=True)
        fh = open(answers_file, "a")
    for i in range(len(val)):
        b = dict_to_device(val[i], device)
        cut = prompt_cut(b["input_ids"], colon_id)
        out = model.generate(b["input_ids"][:, :cut], images=b["images_clip"], attention_mask=b["attention_mask"][:, :cut],
                             max_new_tokens=max_new_tokens, mask_images=b.get("mask_images"),
                             image_token_types=b.get("image_token_types"), image_token_lengths=b.get("image_token_lengths"),
                             region_masks=b.get("region_masks"), valid_region_masks_bool=b.get("valid_region_masks_bool"))
        outs.append(out)
        if fh is not None:
            new_ids = torch.as_tensor(out)[:, cut:]
            rec = {"question_id": i, "image_path": (b.get("image_paths") or [None])[0],
                   "prompt": (b.get("questions_list") or [[None]])[0], "gt": (b.get("gts_list") or [[None]])[0]}
            if tokenizer is not None and hasattr(tokenizer, "batch_decode"):
                rec["text"] = tokenizer.batch_decode(new_ids, skip_special_tokens=True)[0].strip()
            else:
                rec["output_ids"] = new_ids[0].tolist()
            fh.write(json.dumps(rec) + "\n"); fh.flush()
    if fh is not None:
        fh.close()
    return outs


def main(argv=None):
    args = parse_args(argv)
    device = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")))
    torch.cuda.set_device(device)
    torch.manual_seed(args.seed)
    cfg, model = build_model(args, device)
    if args.dataset == "synthetic":
        val = SyntheticDataset(cfg, 1, args.n_samples, args.seed + 7, args.model_size == "tiny")
    else:
        mod, _, fn = args.dataset.partition(":")
        val = getattr(importlib.import_module(mod), fn)(args, cfg)
    out = {}
    if args.eval_vqa:
        out["vqa_output_ids"] = run_vqa(val, model, device, args.max_new_tokens, args.colon_token_id, args.answers_file,
                                        getattr(val, "tokenizer", None))
    if args.eval_seg:
        out["miou"], out["mdice"], out["per_modality"] = validate_seg(val, model, device, args.max_new_tokens, args.colon_token_id)
    return out


if __name__ == "__main__":
    main()
