"""Surface walk of the training driver — NOT a copy of it.

The reference's `train_ds_medplib.py` is the script every `scripts/train_stage*.sh` launches.  A maintainer switching to this build
keeps that script and changes two import lines (`import medplib_amd.engine as deepspeed`,
`from medplib_amd.peft_compat import LoraConfig, get_peft_model`); `model.*`, `datasets` and `utils.utils` resolve to this
repository's packages of the same names.  This module proves that claim without carrying the script: it is a harness, written
for this build, that issues the SAME API CALLS in the same order against the same import faces — one small stage function per
group of calls, listed in `STAGES` with the reference lines each group stands for — and accepts the reference's command line
(`FLAG_TABLE`: names and defaults of train_ds_medplib.py:28-138, kept as data so a flag diff against the reference is one
comparison).  `tests/test_gpu_surface.py` drives it for the dense, LoRA and LoRA + MoE configurations.

Additions of this build (optional flags at the end of the table): `--dataset synthetic` (seeded batches of SURVEY §8d — no dataset
or tokenizer files exist on the build / GPU boxes), `--steps_per_epoch`, `--val_samples`, `--tokenizer_path`, `--seed`."""
import argparse
import itertools
import json
import math
import os
import shutil
import sys
import time
import types
from functools import partial
from pathlib import Path

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# ---- the reference driver's import block, as a maintainer leaves it after the switch (train_ds_medplib.py:11-26)
import medplib_amd.engine as deepspeed                                                              # noqa: E402  was: import deepspeed
from medplib_amd.peft_compat import LoraConfig, get_peft_model                                      # noqa: E402  was: from peft import ...
from model.LISA import LISAForCausalLM                                                              # noqa: E402  unchanged
from model.MedPLIB import MedPLIBForCausalLM                                                        # noqa: E402  unchanged
from datasets import DataCollatorForSupervisedDataset, ICLLazySupervisedDataset, LazySupervisedDataset   # noqa: E402  unchanged
from utils.utils import (ADD_OTHERS_TOKENS, DEFAULT_IM_END_TOKEN, DEFAULT_IM_START_TOKEN, AverageMeter, ProgressMeter,   # noqa: E402
                         Summary, dict_to_cuda, intersectionAndUnionGPU)                            # unchanged

ON, OFF = "store_true/default-on", "store_true/default-off"
FLAG_TABLE = (
    # name, default, kind (a type, ON / OFF, or a tuple of choices)            reference CLI, train_ds_medplib.py:28-138
    ("local_rank", 0, int), ("version", "/path/to/llava-v1.5-7b", str), ("vis_save_path", "./vis_output", str),
    ("pretrain_mm_mlp_adapter", None, str), ("precision", "bf16", ("fp32", "bf16", "fp16")), ("sam_img_size", 256, int),
    ("model_max_length", 512, int), ("vision_tower", "openai/clip-vit-large-patch14", str),
    ("vision_pretrained", "PATH_TO_SAM_ViT-H", str), ("sft_modules", "lm_head,embed_tokens,mask_decoder,text_hidden_fcs", str),
    ("lora_r", 8, int), ("lora_alpha", 16, int), ("lora_dropout", 0.05, float), ("lora_target_modules", "q_proj,v_proj", str),
    ("image_folder", "/path/to/SAMed2D_v1", str), ("image_aspect_ratio", "pad", str), ("is_multimodal", True, bool),
    ("data_path", "/path/to/xxx.json", str), ("val_data_path", "/path/to/xxx.json", str),
    ("icl_enable", False, OFF), ("icl_mask_mode", "overlay", ("overlay", "separate")), ("icl_mask_encoder", False, OFF),
    ("mask_encoder_token_count", 64, int), ("mm_token_compress", False, OFF), ("mm_compressed_token_count", 256, int),
    ("log_base_dir", "./runs", str), ("exp_name", "lisa", str), ("epochs", 10, int), ("batch_size", 2, int),
    ("grad_accumulation_steps", 10, int), ("val_batch_size", 1, int), ("workers", 4, int), ("lr", 0.0003, float),
    ("ce_loss_weight", 1.0, float), ("dice_loss_weight", 0.5, float), ("bce_loss_weight", 2.0, float),
    ("iou_loss_weight", 2.0, float), ("focal_loss_weight", 2.0, float), ("beta1", 0.9, float), ("beta2", 0.95, float),
    ("no_eval", False, OFF), ("eval_only", False, OFF), ("out_dim", 256, int), ("resume", "", str), ("print_freq", 1, int),
    ("save_steps", 10, int), ("start_epoch", 0, int), ("gradient_checkpointing", True, ON), ("train_mask_decoder", False, OFF),
    ("use_mm_start_end", True, ON), ("auto_resume", True, ON), ("conv_type", "llava_v1", ("llava_v1", "llava_llama_2")),
    ("region_fea_adapter", False, OFF), ("region_geo_sampler", False, OFF), ("max_sample_point", 512, int),
    ("sampler_pooler_mode", "max", str), ("moe_enable", False, bool),
    ("moe_mode", "second_half", ("first_half", "second_half", "sparse", "dense")), ("num_experts", 3, int),
    ("top_k_experts", 2, int), ("capacity_factor", 1.0, float), ("use_residual", False, bool),
    ("router_aux_loss_coef", 0.01, float), ("eval_capacity_factor", 2.0, float), ("moe_layers_idx", None, str),
    ("min_capacity", 0, int), ("ep_size", 1, int), ("expert_pretrained_path", None, str), ("finetune_moe", False, bool),
    ("load_in_8bit", False, OFF), ("load_in_4bit", False, OFF), ("num_classes_per_sample", 3, int), ("exclude_val", False, OFF),
    # ---- this build's additions
    ("dataset", "json", ("json", "synthetic")), ("steps_per_epoch", 0, int), ("val_samples", 0, int), ("tokenizer_path", "", str),
    ("seed", 42, int),
)


def parse_args(argv):
    """argparse parser generated from FLAG_TABLE (`type=bool` flags keep argparse's any-non-empty-string-is-True reading, which is
    what `--moe_enable True` in the shipped scripts relies on)."""
    ap = argparse.ArgumentParser(description="MedPLIB training: surface walk of the reference driver")
    for name, default, kind in FLAG_TABLE:
        if kind in (ON, OFF):
            ap.add_argument("--" + name, action="store_true", default=default)
        elif isinstance(kind, tuple):
            ap.add_argument("--" + name, default=default, type=str, choices=list(kind))
        else:
            ap.add_argument("--" + name, default=default, type=kind)
    return ap.parse_args(argv)


class Run(types.SimpleNamespace):
    """What the stages hand to each other."""


# ------------------------------------------------------------------------------------------------------------------ stages
def stage_process_setup(run):
    a = run.args
    a.log_dir = os.path.join(a.log_base_dir, a.exp_name)
    a.local_rank = int(os.environ.get("LOCAL_RANK", a.local_rank))
    torch.cuda.set_device(a.local_rank)
    deepspeed.init_distributed(dist_backend="nccl")
    torch.manual_seed(a.seed)
    run.world = int(os.environ.get("WORLD_SIZE", "1"))
    run.rank0 = a.local_rank == 0
    if run.rank0:
        Path(a.log_dir).mkdir(parents=True, exist_ok=True)
    if isinstance(a.moe_layers_idx, str):
        a.moe_layers_idx = list(map(int, a.moe_layers_idx.split(",")))
    a.num_experts = [a.num_experts]                      # the model classes read a list (medplib_moe_llama.py:597)


def stage_tokenizer(run):
    """AutoTokenizer + the added tokens, in the reference's order (<SEG> etc. first, then the image brackets); without tokenizer
    files (synthetic runs) the checkpoint's own ids stand in: vocabulary unchanged, <SEG> = config.seg_token_idx."""
    a = run.args
    src = a.tokenizer_path or a.version
    if not (os.path.isdir(src) and any(os.path.exists(os.path.join(src, f)) for f in ("tokenizer.model", "tokenizer.json"))):
        meta = json.loads(Path(a.version, "config.json").read_text())
        run.tokenizer, run.n_tokens, a.seg_token_idx = None, int(meta["vocab_size"]), int(meta.get("seg_token_idx", 32000))
        return
    import transformers
    tok = transformers.AutoTokenizer.from_pretrained(src, cache_dir=None, model_max_length=a.model_max_length, padding_side="right",
                                                     use_fast=False, legacy=True)
    tok.pad_token = tok.unk_token
    for extra in ADD_OTHERS_TOKENS:
        tok.add_tokens(extra, special_tokens=True)
    a.seg_token_idx = tok("<SEG>", add_special_tokens=False).input_ids[0]
    if a.use_mm_start_end:
        tok.add_tokens([DEFAULT_IM_START_TOKEN, DEFAULT_IM_END_TOKEN], special_tokens=True)
    run.tokenizer, run.n_tokens = tok, len(tok)


DTYPES = {"fp32": torch.float32, "bf16": torch.bfloat16, "fp16": torch.half}


def stage_open_checkpoint(run):
    a = run.args
    run.dtype = DTYPES[a.precision]
    cls = MedPLIBForCausalLM if a.moe_enable else LISAForCausalLM
    run.model = cls.from_pretrained(a.version, torch_dtype=run.dtype, low_cpu_mem_usage=True, ignore_mismatched_sizes=True, **vars(a))
    if run.tokenizer is not None:
        for field in ("eos_token_id", "bos_token_id", "pad_token_id"):
            setattr(run.model.config, field, getattr(run.tokenizer, field))
    run.model.enable_input_require_grads()
    run.model.gradient_checkpointing_enable()


def stage_vision_modules(run):
    a, inner = run.args, run.model.get_model()
    inner.initialize_vision_modules(inner.config)
    if not a.eval_only:
        (inner.initialize_bird_modules if a.moe_enable else inner.initialize_lisa_modules)(inner.config)
    run.tower = inner.get_vision_tower()
    run.tower.to(dtype=run.dtype, device=a.local_rank)
    for frozen in itertools.chain(run.tower.parameters(), inner.mm_projector.parameters()):
        frozen.requires_grad = False


NEVER_ADAPTED = ("visual_model", "vision_tower", "mm_projector")


def stage_adapters(run):
    """LoRA targets = every nn.Linear whose name carries one of --lora_target_modules and none of the vision-side prefixes; with
    --lora_r 0 the whole model is frozen instead and --sft_modules alone decides what trains."""
    a = run.args
    if a.lora_r <= 0:
        for _, p in run.model.named_parameters():
            p.requires_grad = False
        return
    wanted = a.lora_target_modules.split(",")
    targets = sorted({name for name, mod in run.model.named_modules()
                      if isinstance(mod, torch.nn.Linear) and not any(s in name for s in NEVER_ADAPTED) and any(w in name for w in wanted)})
    if run.rank0:
        print(f"[walk] {len(targets)} LoRA targets, e.g. {targets[:3]}")
    run.model = get_peft_model(run.model, LoraConfig(r=a.lora_r, lora_alpha=a.lora_alpha, target_modules=targets,
                                                     lora_dropout=a.lora_dropout, bias="none", task_type="CAUSAL_LM"))
    run.model.print_trainable_parameters()


def stage_moe_and_vocabulary(run):
    if run.args.moe_enable:
        run.model.initialize_moe_modules(run.args)
    run.model.resize_token_embeddings(run.n_tokens)


def stage_sft_flags(run):
    wanted = [w for w in run.args.sft_modules.split(",") if w]
    n_train = n_all = 0
    for name, p in run.model.named_parameters():
        if wanted and any(w in name for w in wanted):
            p.requires_grad = True
        n_all += p.numel()
        n_train += p.numel() if p.requires_grad else 0
    if run.rank0:
        print(f"[walk] trainable {n_train} of {n_all} parameters ({100.0 * n_train / max(n_all, 1):.7f} %)")


def stage_data(run):
    """LazySupervisedDataset / ICLLazySupervisedDataset over the JSON files with the tower's image processor, or seeded synthetic
    micro-batches (already collated: the collate function is the identity on a one-item list)."""
    a = run.args
    if a.dataset == "synthetic":
        from medplib_amd.train import SyntheticDataset, synth_batch
        cfg = run.model.config
        tiny = cfg.hidden_size < 1024
        a.steps_per_epoch = a.steps_per_epoch or 4
        run.train_set = SyntheticDataset(cfg, a.batch_size, a.steps_per_epoch * a.grad_accumulation_steps * a.epochs,
                                         a.seed + 1000 * a.local_rank, tiny)
        run.collate, run.micro = (lambda items: items[0]), 1
        run.val_batches = [dict(synth_batch(cfg, 1, a.seed + 7000 + i, tiny=tiny), inference=True) for i in range(a.val_samples)]
        if not run.val_batches:
            a.no_eval = True
        return
    data_args = types.SimpleNamespace(
        image_folder=a.image_folder, image_aspect_ratio=a.image_aspect_ratio, is_multimodal=a.is_multimodal,
        mm_use_im_start_end=a.use_mm_start_end, data_path=a.data_path, icl_mask_mode=a.icl_mask_mode,
        icl_mask_encoder=a.icl_mask_encoder, mask_encoder_token_count=a.mask_encoder_token_count,
        mm_token_compress=a.mm_token_compress, mm_compressed_token_count=a.mm_compressed_token_count,
        image_processor=run.tower.image_processor)
    make = ICLLazySupervisedDataset if a.icl_enable else LazySupervisedDataset
    run.train_set = make(a.data_path, run.tokenizer, data_args, a.sam_img_size)
    per_rank = math.ceil(len(run.train_set) / (a.batch_size * run.world))
    a.steps_per_epoch = math.ceil(per_rank / a.grad_accumulation_steps)
    run.collate, run.micro = partial(DataCollatorForSupervisedDataset), a.batch_size
    run.val_batches = None
    if not a.no_eval:
        assert a.val_batch_size == 1
        val_set = make(a.val_data_path, run.tokenizer, data_args, a.sam_img_size)
        sampler = torch.utils.data.distributed.DistributedSampler(val_set, shuffle=False, drop_last=False) if run.world > 1 else None
        run.val_batches = torch.utils.data.DataLoader(val_set, batch_size=1, shuffle=False, num_workers=a.workers, pin_memory=False,
                                                      sampler=sampler, collate_fn=partial(DataCollatorForSupervisedDataset, inference=True))


def engine_config(a, micro):
    """The ds_config dict of train_ds_medplib.py:383-420 (ZeRO-2 bf16 AdamW, WarmupDecayLR over epochs x steps, warm-up = 1 % of an
    epoch, clip 1.0), built from the flags."""
    sched = {"total_num_steps": a.epochs * a.steps_per_epoch, "warmup_min_lr": 0, "warmup_max_lr": a.lr,
             "warmup_num_steps": int(a.steps_per_epoch * 0.01), "warmup_type": "linear"}
    return {"train_micro_batch_size_per_gpu": micro, "gradient_accumulation_steps": a.grad_accumulation_steps,
            "optimizer": {"type": "AdamW", "params": {"lr": a.lr, "weight_decay": 0.0, "betas": (a.beta1, a.beta2)}},
            "scheduler": {"type": "WarmupDecayLR", "params": sched}, "gradient_clipping": 1.0,
            "fp16": {"enabled": a.precision == "fp16"}, "bf16": {"enabled": a.precision == "bf16"},
            "zero_optimization": {"stage": 2, "contiguous_gradients": True, "overlap_comm": True, "reduce_scatter": True,
                                  "reduce_bucket_size": 5e8, "allgather_bucket_size": 5e8}}


def stage_engine(run):
    a = run.args
    params = run.model.parameters()
    if a.moe_enable and "up_proj" in a.lora_target_modules:       # expert adapters get their own optimizer groups (:422-434)
        params = deepspeed.split_params_into_different_moe_groups_for_optimizer({"params": list(params), "name": "parameters"})
    run.engine, run.optimizer, run.loader, run.scheduler = deepspeed.initialize(
        model=run.model, model_parameters=params, training_data=run.train_set, collate_fn=run.collate,
        config=engine_config(a, run.micro))


def stage_resume(run):
    a = run.args
    candidate = Path(a.log_dir, "ckpt_model")
    if a.auto_resume and not a.resume and candidate.exists():
        a.resume = str(candidate)
    if not a.resume:
        return
    run.engine.load_checkpoint(a.resume)
    tag = Path(a.resume, "latest").read_text().splitlines()[0].strip()
    a.start_epoch = int(tag.replace("global_step", "")) // a.steps_per_epoch
    if run.rank0:
        print(f"[walk] resumed {a.resume} at {tag} (engine.global_steps {run.engine.global_steps}) -> epoch {a.start_epoch}")


def fresh_checkpoint(run, name):
    """save_checkpoint into a directory emptied first by rank 0 (the driver keeps one rolling checkpoint per name)."""
    target = os.path.join(run.args.log_dir, name)
    if run.rank0 and os.path.exists(target):
        shutil.rmtree(target)
    if torch.distributed.is_initialized():
        torch.distributed.barrier()
    run.engine.save_checkpoint(target)


def stage_epochs(run):
    a = run.args
    run.history, run.val_scores = [], []
    if a.eval_only:
        if not run.val_batches:
            raise ValueError("--eval_only needs validation samples (--val_data_path, or --val_samples N with --dataset synthetic)")
        run.val_scores.append(validate(run, 0))
        return
    feed = endless(run.loader)
    for epoch in range(a.start_epoch, a.epochs):
        run.history += train_epoch(run, epoch, feed)
        if not a.no_eval:
            run.val_scores.append(validate(run, epoch))
        fresh_checkpoint(run, "last_ckpt_model")


STAGES = (
    # (stage, the reference lines whose calls it issues)
    (stage_process_setup, "train_ds_medplib.py:181-196"), (stage_tokenizer, ":198-216"), (stage_open_checkpoint, ":218-238"),
    (stage_vision_modules, ":240-259"), (stage_adapters, ":261-306"), (stage_moe_and_vocabulary, ":308-312"),
    (stage_sft_flags, ":315-349"), (stage_data, ":352-381,472-489"), (stage_engine, ":383-448"), (stage_resume, ":452-470"),
    (stage_epochs, ":491-533"),
)


# ------------------------------------------------------------------------------------------------------------------ loops
def endless(loader):
    while True:
        yield from loader


def to_device_in_dtype(batch, precision):
    """dict_to_cuda + the image casts both loops of the reference apply (:586-598, :745-757)."""
    batch = dict_to_cuda(batch)
    want = DTYPES[precision]
    batch["images"] = batch["images"].to(want)
    clip = batch["images_clip"]
    batch["images_clip"] = [c.to(want) for c in clip] if isinstance(clip, list) else clip.to(want)
    return batch


LOSS_METERS = ("loss", "ce_loss", "mask_bce_loss", "mask_dice_loss", "mask_loss", "unscale_mask_bce_loss", "unscale_mask_dice_loss",
               "unscale_mask_loss", "unscale_mask_iou_loss", "unscale_mask_focal_loss")


def train_epoch(run, epoch, feed):
    """steps_per_epoch optimizer steps of grad_accumulation_steps micro-batches each: engine(**batch) -> engine.backward(loss) ->
    engine.step(); ten loss meters + two timers shown through ProgressMeter every print_freq steps (all-reduced when distributed),
    `ckpt_model` rewritten every save_steps steps (:536-700).  Returns the losses it displayed."""
    a, eng = run.args, run.engine
    clock = {"Time": AverageMeter("Time", ":6.2f"), "Data": AverageMeter("Data", ":6.2f")}
    meters = {k: AverageMeter(k, ":.4f") for k in LOSS_METERS}
    board = ProgressMeter(a.steps_per_epoch, list(clock.values()) + list(meters.values()), prefix=f"Epoch: [{epoch}]")
    eng.train()
    shown, tick = [], time.time()
    # a checkpoint taken inside an epoch (`ckpt_model`, every save_steps): the epoch continues where it stopped — the optimizer steps
    # already taken are not repeated and their micro-batches are drawn and dropped, so the step count, the LR schedule and the data
    # position line up with an uninterrupted run (:567-578)
    done = eng.global_steps % a.steps_per_epoch
    if done and run.rank0:
        print(f"[walk] skipping first {done} steps, global step is {eng.global_steps}", flush=True)
    for _ in range(done * a.grad_accumulation_steps):
        next(feed)
    for step in range(done, a.steps_per_epoch):
        for _ in range(a.grad_accumulation_steps):
            batch = next(feed)
            clock["Data"].update(time.time() - tick)
            batch = to_device_in_dtype(batch, a.precision)
            out = eng(**batch)
            n = batch["images"].size(0)
            for k, m in meters.items():
                if k in ("loss", "ce_loss") or batch["seg_flag"]:
                    m.update(out[k].item(), n)
            eng.backward(out["loss"])
            eng.step()
        clock["Time"].update(time.time() - tick)
        tick = time.time()
        if step % a.print_freq == 0:
            if run.world > 1:
                for m in itertools.chain(clock.values(), meters.values()):
                    m.all_reduce()
            shown.append(meters["loss"].avg)
            if run.rank0:
                board.display(step + 1)
                print(f"[walk] global step {eng.global_steps} lr {run.scheduler.get_last_lr()[0]:.3e}", flush=True)
            for m in itertools.chain(clock.values(), meters.values()):
                m.reset()
        if step != 0 and step % a.save_steps == 0:
            fresh_checkpoint(run, "ckpt_model")
    return shown


@torch.no_grad()
def validate(run, epoch):
    """One pass over the validation samples in inference mode ({pred_masks, gt_masks}): sigmoid > 0.1, class counts through
    intersectionAndUnionGPU (K = 2), the "no-object target" rule, per-sample IoU / Dice, five SUM meters reduced over the ranks;
    gIoU = mean acc_iou[1], cIoU = (sum I / sum U)[1] (:721-800).  -> (giou, ciou, miou, mdice)."""
    a, eng = run.args, run.engine
    names = ("Intersec", "Union", "gIoU", "IoU", "Dice")
    m = {k: AverageMeter(k, ":6.3f", Summary.SUM) for k in names}
    eng.eval()
    for batch in run.val_batches:
        out = eng(**to_device_in_dtype(dict(batch), a.precision))
        target = out["gt_masks"][0].int().unsqueeze(0)
        guess = (torch.sigmoid(out["pred_masks"][0].float()) > 0.1).int()
        inter = union = acc = 0.0
        for g, p in zip(target, guess):
            g, p = g.unsqueeze(0), p.unsqueeze(0)
            i_k, u_k, _ = intersectionAndUnionGPU(p.contiguous().clone(), g.contiguous(), 2, ignore_index=255)
            ratio = i_k / (u_k + 1e-5)
            ratio[u_k == 0] += 1.0
            inter, union, acc = inter + i_k, union + u_k, acc + ratio
            both, either = int((p.bool() & g.bool()).sum()), int((p.bool() | g.bool()).sum())
            iou = both / either if either else 0.0
        k = target.shape[0]
        m["Intersec"].update(inter.cpu().numpy()); m["Union"].update(union.cpu().numpy()); m["gIoU"].update(acc.cpu().numpy() / k, n=k)
        m["IoU"].update(iou); m["Dice"].update(2 * iou / (1 + iou))
    for meter in m.values():
        meter.all_reduce()
    ciou = (m["Intersec"].sum / (m["Union"].sum + 1e-10))[1]
    giou = m["gIoU"].avg[1]
    if run.rank0:
        print("giou: {:.6f}, ciou: {:.6f}".format(giou, ciou))
        print("miou: {:.6f}, mDice: {:.6f}".format(m["IoU"].avg, m["Dice"].avg))
    eng.train()
    return float(giou), float(ciou), float(m["IoU"].avg), float(m["Dice"].avg)


def main(argv):
    run = Run(args=parse_args(argv))
    for stage, _ in STAGES:
        stage(run)
    main.last_run = run                                     # tests read run.val_scores here
    return run.history


if __name__ == "__main__":
    main(sys.argv[1:])
