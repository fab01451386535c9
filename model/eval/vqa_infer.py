"""Surface walk of the inference driver — the call shapes of the reference's `model/eval/vqa_infer.py` through `model.MedPLIB` /
`model.LISA`, written for this build (no statement of the reference script is carried).

What the reference driver does with the model, in order, and where this harness does the same through the same import faces:
`Cls.from_pretrained(version, torch_dtype=, low_cpu_mem_usage=True, ignore_mismatched_sizes=True, **vars(args), test_only=True)`
(:244-256) -> token ids on `model.config` + `resize_token_embeddings(len(tokenizer))` (:258-261) -> `get_model().get_vision_tower()`
(`.image_processor` for the datasets, :264,285) -> `model.to(dtype=, device=)` (:267) -> every parameter frozen (:270-272) ->
`datasets.{ICL,}LazySupervisedDataset` + `DataLoader(collate_fn=partial(DataCollatorForSupervisedDataset, inference=True))`
(:288-312) -> `model.eval()` -> `--eval_seg`: per sample `model.evaluate(images_clip, images, input_ids, resize_list, label_list,
max_new_tokens=, tokenizer=, attention_mask=, mask_images=, image_token_types=, image_token_lengths=)` (:528-540), threshold 0.1,
IoU / Dice meters, per-modality table (:565-633) | `--eval_vqa`: per sample `model.generate(input_ids, images=, attention_mask=,
mask_images=, image_token_types=, do_sample=, temperature=, top_p=, num_beams=, max_new_tokens=, use_cache=True)` (:430-442), one
JSON line per answer (:470-480).  The prompt is cut after the last ':' token (id 29901 with the Llama tokenizer, :426-428,521-523).

`FLAG_TABLE` holds the reference's command line as data.  Additions of this build: `--dataset synthetic` (seeded single-sample
batches; no dataset / tokenizer files exist on the build and GPU boxes), `--n_samples`, `--max_new_tokens`, `--colon_token_id`,
`--tokenizer_path`."""
import argparse
import json
import os
import sys
import types
from functools import partial
from pathlib import Path

import torch

ROOT = str(Path(__file__).resolve().parents[2])
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import medplib_amd.engine as deepspeed                                                              # noqa: E402  was: import deepspeed
from model.LISA import LISAForCausalLM                                                              # noqa: E402  unchanged
from model.MedPLIB import MedPLIBForCausalLM                                                        # noqa: E402  unchanged
from datasets import DataCollatorForSupervisedDataset, ICLLazySupervisedDataset, LazySupervisedDataset   # noqa: E402  unchanged
from utils.utils import AverageMeter, Summary, dict_to_cuda                                         # noqa: E402  unchanged

OFF, ON = "off-by-default switch", "on-by-default switch"
FLAG_TABLE = (
    # reference CLI, model/eval/vqa_infer.py:33-157: (name, default, kind)
    ("local_rank", 0, int), ("version", "/root/huggingface_models/llava-v1.5-7b", str), ("vis_save_path", "./vis_output", str),
    ("pretrain_mm_mlp_adapter", None, str), ("precision", "bf16", ("fp32", "bf16", "fp16")), ("sam_img_size", 256, int),
    ("model_max_length", 2048, int), ("vision_tower", "openai/clip-vit-large-patch14", str), ("image_folder", "", str),
    ("image_aspect_ratio", "pad", str), ("is_multimodal", True, ON), ("val_data_path", "", str), ("answer_type", "closed", str),
    ("icl_enable", False, OFF), ("icl_mask_mode", "overlay", ("overlay", "separate")), ("icl_mask_encoder", False, OFF),
    ("mask_encoder_token_count", 64, int), ("mm_token_compress", False, OFF), ("mm_compressed_token_count", 256, int),
    ("val_batch_size", 1, int), ("workers", 1, int), ("ce_loss_weight", 1.0, float), ("dice_loss_weight", 0.5, float),
    ("bce_loss_weight", 2.0, float), ("iou_loss_weight", 2.0, float), ("focal_loss_weight", 2.0, float),
    ("vision_pretrained", "PATH_TO_SAM_ViT-H", str), ("out_dim", 256, int), ("use_mm_start_end", True, ON), ("eval_seg", False, OFF),
    ("eval_vqa", False, OFF), ("temperature", 0.0, float), ("top_p", None, float), ("num_beams", 1, int), ("cpu_only", False, OFF),
    ("vis_mask", False, OFF), ("num-chunks", 1, int), ("chunk-idx", 0, int), ("answers-file", "", str),
    ("region_fea_adapter", False, OFF), ("region_geo_sampler", False, OFF), ("max_sample_point", 512, int),
    ("sampler_pooler_mode", "max", str), ("moe_enable", False, OFF),
    ("moe_mode", "second_half", ("first_half", "second_half", "sparse", "dense")), ("num_experts", 3, int), ("top_k_experts", 2, int),
    ("capacity_factor", 1.0, float), ("use_residual", False, OFF), ("router_aux_loss_coef", 0.01, float),
    ("eval_capacity_factor", 2.0, float), ("moe_layers_idx", None, str), ("min_capacity", 0, int), ("ep_size", 1, int),
    ("expert_pretrained_path", None, str), ("return_gating_logit", False, OFF),
    # ---- this build's additions
    ("dataset", "json", ("json", "synthetic")), ("n_samples", 4, int), ("max_new_tokens", 1024, int), ("colon_token_id", 29901, int),
    ("tokenizer_path", "", str), ("seed", 42, int),
)


def parse_args(argv):
    ap = argparse.ArgumentParser(description="MedPLIB inference: surface walk of the reference driver")
    for name, default, kind in FLAG_TABLE:
        if kind in (ON, OFF):
            ap.add_argument("--" + name, action="store_true", default=default)
        elif isinstance(kind, tuple):
            ap.add_argument("--" + name, default=default, type=str, choices=list(kind))
        else:
            ap.add_argument("--" + name, default=default, type=kind)
    return ap.parse_args(argv)


def open_model(args, tokenizer):
    """Construction half of the walk.  -> (model, vision tower)."""
    if args.cpu_only:
        raise NotImplementedError("--cpu_only: this build has no CPU path (the HIP library is the only implementation)")
    if args.precision != "bf16":
        raise ValueError("this build computes in bf16 (the reference's default --precision)")
    if isinstance(args.moe_layers_idx, str):
        args.moe_layers_idx = [int(t) for t in args.moe_layers_idx.split(",")]
    if not isinstance(args.num_experts, list):
        args.num_experts = [args.num_experts]
    kwargs = dict(vars(args), test_only=True)
    cls = MedPLIBForCausalLM if args.moe_enable else LISAForCausalLM
    model = cls.from_pretrained(args.version, torch_dtype=torch.bfloat16, low_cpu_mem_usage=True, ignore_mismatched_sizes=True, **kwargs)
    if tokenizer is not None:
        for field in ("eos_token_id", "bos_token_id", "pad_token_id"):
            setattr(model.config, field, getattr(tokenizer, field))
        model.resize_token_embeddings(len(tokenizer))
    tower = model.get_model().get_vision_tower()
    model.to(dtype=torch.bfloat16, device=args.local_rank)
    for _, p in model.named_parameters():
        p.requires_grad = False
    return model, tower


def open_data(args, model, tower, tokenizer):
    """-> iterable of collated single-sample batches (inference=True)."""
    if args.dataset == "synthetic":
        from medplib_amd.train import synth_batch
        cfg = model.config
        out = []
        for i in range(args.n_samples):
            b = synth_batch(cfg, 1, args.seed + 7 + i, tiny=cfg.hidden_size < 1024)
            b["input_ids"][:, 55] = args.colon_token_id          # the ':' that closes "ASSISTANT:" — the prompt ends here
            b["inference"] = True
            b["image_paths"] = [f"synthetic_{i:03d}.png"]
            out.append(b)
        return out
    data_args = types.SimpleNamespace(
        image_folder=args.image_folder, image_aspect_ratio=args.image_aspect_ratio, is_multimodal=args.is_multimodal,
        mm_use_im_start_end=args.use_mm_start_end, icl_mask_mode=args.icl_mask_mode, icl_mask_encoder=args.icl_mask_encoder,
        mask_encoder_token_count=args.mask_encoder_token_count, mm_token_compress=args.mm_token_compress,
        mm_compressed_token_count=args.mm_compressed_token_count, image_processor=tower.image_processor)
    make = ICLLazySupervisedDataset if args.icl_enable else LazySupervisedDataset
    val = make(args.val_data_path, tokenizer, data_args, args.sam_img_size)
    if args.eval_vqa:                                            # --num-chunks / --chunk-idx: contiguous slices of the sample list
        per = -(-len(val) // args.num_chunks)
        val = torch.utils.data.Subset(val, range(len(val))[args.chunk_idx * per:(args.chunk_idx + 1) * per])
    assert args.val_batch_size == 1
    return torch.utils.data.DataLoader(val, batch_size=1, shuffle=False, num_workers=args.workers, pin_memory=False, drop_last=False,
                                       collate_fn=partial(DataCollatorForSupervisedDataset, inference=True))


def _staged(batch):
    batch = dict_to_cuda(batch)
    clip = batch["images_clip"]
    batch["images"] = batch["images"].bfloat16()
    batch["images_clip"] = [c.bfloat16() for c in clip] if isinstance(clip, list) else clip.bfloat16()
    return batch


def _prompt_part(batch, colon_id):
    """(input_ids, attention_mask) up to and including the last ':' of the row."""
    ids = torch.as_tensor(batch["input_ids"])
    hits = (ids == colon_id).nonzero(as_tuple=True)[1]
    if hits.numel() == 0:
        raise ValueError(f"no token {colon_id} (':' of 'ASSISTANT:') in the prompt; pass --colon_token_id for another tokenizer")
    end = int(hits[-1]) + 1
    return ids[:, :end], torch.as_tensor(batch["attention_mask"])[:, :end]


@torch.no_grad()
def validate_seg(val, model, args, tokenizer):
    """-> (mIoU, mDice, per-modality means).  One `model.evaluate` per sample in the reference's argument order."""
    iou_meter, dice_meter = AverageMeter("IoU", ":6.3f", Summary.SUM), AverageMeter("Dice", ":6.3f", Summary.SUM)
    by_modality = {}
    for batch in val:
        batch = _staged(batch)
        input_ids, attention_mask = _prompt_part(batch, args.colon_token_id)
        output_ids, pred_masks = model.evaluate(
            batch["images_clip"], batch["images"], input_ids, batch["resize_list"], batch["label_list"],
            max_new_tokens=args.max_new_tokens, tokenizer=tokenizer, attention_mask=attention_mask,
            mask_images=batch.get("mask_images", None), image_token_types=batch.get("image_token_types", None),
            image_token_lengths=batch.get("image_token_lengths", None))
        iou = 0.0
        if len(pred_masks) > 0:
            target = batch["masks_list"][0].to(pred_masks[0].device).bool().reshape(-1)
            guess = (torch.sigmoid(pred_masks[0].float()) > 0.1).reshape(-1)
            either = int((guess | target).sum())
            iou = int((guess & target).sum()) / either if either else 0.0
        dice = 2 * iou / (1 + iou)
        iou_meter.update(iou); dice_meter.update(dice)
        modality = os.path.basename(batch["image_paths"][0]).split("_")[0]
        slot = by_modality.setdefault(modality, {"iou": [], "dice": []})
        slot["iou"].append(iou); slot["dice"].append(dice)
    print("miou: {:.6f}, mDice: {:.6f}".format(iou_meter.avg, dice_meter.avg))
    table = {m: {k: round(sum(v) / len(v), 6) for k, v in d.items()} for m, d in by_modality.items()}
    print(table)
    return iou_meter.avg, dice_meter.avg, table


@torch.no_grad()
def validate_vqa(val, model, args, tokenizer):
    """-> list of new-token id lists.  One `model.generate` per sample with the reference's keyword set; answers appended to
    --answers-file as JSON lines (`text` when a tokenizer can decode, the raw new ids otherwise)."""
    sink = None
    if args.answers_file:
        Path(args.answers_file).resolve().parent.mkdir(parents=True, exist_ok=True)
        sink = open(args.answers_file, "a")
    answers = []
    for idx, batch in enumerate(val):
        batch = _staged(batch)
        input_ids, attention_mask = _prompt_part(batch, args.colon_token_id)
        output_ids = model.generate(
            input_ids, images=batch["images_clip"], attention_mask=attention_mask, mask_images=batch.get("mask_images", None),
            image_token_types=batch.get("image_token_types", None), do_sample=True if args.temperature > 0 else False,
            temperature=args.temperature, top_p=args.top_p, num_beams=args.num_beams, max_new_tokens=args.max_new_tokens, use_cache=True)
        new = torch.as_tensor(output_ids)[:, input_ids.shape[1]:]
        answers.append(new[0].tolist())
        if sink is not None:
            record = {"question_id": idx, "image_path": batch["image_paths"][0], "answer_type": args.answer_type,
                      "prompt": (batch.get("questions_list") or [[None]])[0], "gt": (batch.get("gts_list") or [[None]])[0]}
            if tokenizer is not None:
                record["text"] = tokenizer.batch_decode(new, skip_special_tokens=True)[0].strip()
            else:
                record["output_ids"] = answers[-1]
            sink.write(json.dumps(record) + "\n")
            sink.flush()
    if sink is not None:
        sink.close()
    return answers


def main(argv):
    args = parse_args(argv)
    args.local_rank = int(os.environ.get("LOCAL_RANK", args.local_rank))
    torch.cuda.set_device(args.local_rank)
    deepspeed.init_distributed(dist_backend="nccl")
    torch.manual_seed(args.seed)
    tokenizer = None
    src = args.tokenizer_path or args.version
    if os.path.isdir(src) and any(os.path.exists(os.path.join(src, f)) for f in ("tokenizer.model", "tokenizer.json")):
        import transformers
        tokenizer = transformers.AutoTokenizer.from_pretrained(src, cache_dir=None, model_max_length=args.model_max_length,
                                                               padding_side="right", use_fast=False, legacy=True)
        args.seg_token_idx = tokenizer("<SEG>", add_special_tokens=False).input_ids[0]
    else:
        args.seg_token_idx = int(json.loads(Path(args.version, "config.json").read_text()).get("seg_token_idx", 32000))
    model, tower = open_model(args, tokenizer)
    val = open_data(args, model, tower, tokenizer)
    model.eval()
    out = {}
    if args.eval_seg:
        out["miou"], out["mdice"], out["per_modality"] = validate_seg(val, model, args, tokenizer)
    if args.eval_vqa:
        out["answers"] = validate_vqa(val, model, args, tokenizer)
    return out


if __name__ == "__main__":
    main(sys.argv[1:])
