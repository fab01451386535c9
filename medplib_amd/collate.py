"""Batch assembly with the reference collator's contract (datasets/DataCollatorForSupervisedDataset.py:11-138):
what `model_forward(**batch)` consumes.  The reference's own collator can be used unchanged with this build's models; this mirror
exists so a training loop does not need the reference checkout (`engine.initialize(..., collate_fn=collate)`).

Per-sample dicts carry: input_ids / labels (1-D int64), tokenizer (pad_token_id, model_max_length), masks / label / resize lists,
region_masks, image_sam [3,H,W], image_clip ([3,h,w] or [n_img,3,h,w] for ICL), conversations, and the optional ICL / bookkeeping
fields.  Host-side list handling only — no kernels."""
from typing import Dict, Sequence

import torch

IGNORE_INDEX = -100


def _pad(seqs, value):
    n = max(int(s.shape[0]) for s in seqs)
    out = torch.full((len(seqs), n), value, dtype=seqs[0].dtype)
    for i, s in enumerate(seqs):
        out[i, : s.shape[0]] = s
    return out


def collate(samples: Sequence[Dict], inference: bool = False) -> Dict:
    tok = samples[0]["tokenizer"]
    limit = tok.model_max_length
    ids = _pad([s["input_ids"] for s in samples], tok.pad_token_id)[:, :limit]
    labels = _pad([s["labels"] for s in samples], IGNORE_INDEX)[:, :limit]
    out = {"input_ids": ids, "labels": labels, "attention_mask": ids.ne(tok.pad_token_id)}

    # segmentation targets: flat over the batch, with per-sample validity lists (expand_embedding, MedPLIB.py:292-308)
    has_masks = any(len(s["masks"]) > 0 for s in samples)
    masks, label_list, resize_list, valid = [], [], [], []
    if has_masks:
        for s in samples:
            if len(s["masks"]) > 0:
                masks += list(s["masks"]); label_list += list(s["label"]); resize_list += list(s["resize"])
                valid.append([True] * len(s["masks"]))
            else:
                valid.append([])
    # region prompts
    region_masks, valid_region = [], []
    rp_flag = any(len(s["region_masks"]) > 0 for s in samples)
    if rp_flag:
        for s in samples:
            if len(s["region_masks"]) > 0:
                region_masks += list(s["region_masks"])
                valid_region.append([torch.ones(1).bool()] * len(s["region_masks"]))
            else:
                valid_region.append([torch.zeros(1).bool()])

    conversations, offsets = [], [0]
    for s in samples:
        conversations += list(s.get("conversations"))
        offsets.append(len(conversations))
    clips = [s.get("image_clip") for s in samples]
    images_clip = clips if (clips and clips[0].dim() == 4) else torch.stack(clips, 0)       # list = several images per sample (ICL)
    out.update({
        "image_paths": [s.get("image_path") for s in samples],
        "icl_image_paths": [s.get("icl_image_paths", []) for s in samples],
        "icl_image_counts": [s.get("icl_image_count", 1) for s in samples],
        "mask_images": [s["mask_images"] for s in samples if s.get("mask_images", torch.empty(0)).numel() > 0],
        "image_token_types": [s.get("image_token_types", ["image"]) for s in samples],
        "image_token_lengths": [s.get("image_token_lengths", []) for s in samples],
        "images": torch.stack([s.get("image_sam") for s in samples], 0),
        "images_clip": images_clip,
        "masks_list": masks, "label_list": label_list, "resize_list": resize_list,
        "offset": torch.LongTensor(offsets),
        "questions_list": [s.get("question") for s in samples], "gts_list": [s.get("gt") for s in samples],
        "sampled_classes_list": [s.get("sampled_classes") for s in samples], "conversation_list": conversations,
        "seg_flag": has_masks, "valid_mask_bool": valid, "inference": inference,
        "answer_type_list": [s.get("answer_type") for s in samples],
        "rp_flag": rp_flag, "region_masks": region_masks, "valid_region_masks_bool": valid_region,
    })
    return out
