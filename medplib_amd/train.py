"""Training entry point with the control flow of the reference's `train_ds_medplib.py` (main :181-533, train :536-700, validate
:721-800): build the model from the same flags, `initialize()` the engine from the same `ds_config` shape (:383-420), resume from
`<log_dir>/ckpt_model/latest` (:453-470), loop `steps_per_epoch x grad_accumulation_steps` micro-batches per epoch with
`engine(**batch) / backward / step`, all-reduce the meters every `print_freq` steps, save every `save_steps`, validate per epoch.

`--dataset` names either `synthetic` (the seeded generator of SURVEY §8d, used by the tests and the benchmark) or
`package.module:factory`, a callable `(args, cfg) -> (train, val)` of batch-indexed datasets (`data[i]` = one collated
micro-batch); `medplib_amd.dataset:from_args` is the one for the reference's JSON data files (`--data_path --image_folder
--tokenizer_path`, SURVEY §8f rank 3: v1 prompts, target masking, device-side image preprocessing, ICL assembly).

One deliberate difference: the reference calls `.item()` on ten loss tensors after every micro-batch, i.e. synchronises the host
with the GPU each step; here the meters keep device tensors and are read once per `print_freq` steps, so the host keeps running
ahead of the device."""
import argparse
import importlib
import os
import sys
import time

import torch
import torch.distributed as dist

from . import engine as E
from . import metrics
from .model.config import MedPLIBConfig
from .model.medplib import LOSS_KEYS, LISAForCausalLM, MedPLIBForCausalLM


def parse_args(argv=None):
    p = argparse.ArgumentParser(description="MedPLIB stage-III style training on MI355X")
    # names and defaults follow train_ds_medplib.py:41-138 where the flag exists there
    p.add_argument("--local_rank", default=0, type=int)
    p.add_argument("--version", default="", help="HF-layout state dict (.pt / .bin) to start from; empty = seeded random init")
    p.add_argument("--vision_pretrained", default="", help="SAM-Med2D checkpoint (torch.load(...)['model'] layout)")
    p.add_argument("--precision", default="bf16", choices=["bf16"])
    p.add_argument("--model_size", default="7b", choices=["7b", "tiny"])
    p.add_argument("--dataset", default="synthetic")
    p.add_argument("--data_path", default="", help="training JSON (with --dataset medplib_amd.dataset:from_args)")
    p.add_argument("--val_data_path", default="")
    p.add_argument("--image_folder", default="")
    p.add_argument("--tokenizer_path", default="", help="directory with the Llama tokenizer files")
    p.add_argument("--model_max_length", default=512, type=int)
    p.add_argument("--icl_enable", action="store_true", default=False)
    p.add_argument("--icl_mask_mode", default="overlay", choices=["overlay", "separate"])
    p.add_argument("--icl_mask_encoder", action="store_true", default=False)
    p.add_argument("--log_dir", default="./runs/medplib")
    p.add_argument("--epochs", default=1, type=int)
    p.add_argument("--steps_per_epoch", default=10, type=int)
    p.add_argument("--batch_size", default=8, type=int, help="micro-batch per GPU")
    p.add_argument("--grad_accumulation_steps", default=1, type=int)
    p.add_argument("--lr", default=3e-4, type=float)
    p.add_argument("--beta1", default=0.9, type=float)
    p.add_argument("--beta2", default=0.95, type=float)
    p.add_argument("--ce_loss_weight", default=1.0, type=float)
    p.add_argument("--dice_loss_weight", default=0.5, type=float)
    p.add_argument("--bce_loss_weight", default=2.0, type=float)
    p.add_argument("--iou_loss_weight", default=2.0, type=float)
    p.add_argument("--focal_loss_weight", default=2.0, type=float)
    p.add_argument("--print_freq", default=1, type=int)
    p.add_argument("--save_steps", default=10, type=int)
    p.add_argument("--no_eval", action="store_true", default=False)
    p.add_argument("--eval_only", action="store_true", default=False)
    p.add_argument("--auto_resume", action="store_true", default=True)
    p.add_argument("--resume", default="", type=str)
    p.add_argument("--train_mask_decoder", action="store_true", default=True)
    p.add_argument("--lisa", action="store_true", help="dense LISAForCausalLM instead of the MoE class")
    p.add_argument("--moe_enable", type=lambda s: s.lower() in ("1", "true"), default=True)
    p.add_argument("--num_experts", type=int, default=2)
    p.add_argument("--top_k_experts", type=int, default=1)
    p.add_argument("--capacity_factor", type=float, default=1.5)
    p.add_argument("--eval_capacity_factor", type=float, default=2.0)
    p.add_argument("--min_capacity", type=int, default=0)
    p.add_argument("--lora_r", default=0, type=int)                     # train_ds_medplib.py:57-60 (reference default 8; 0 = LoRA off)
    p.add_argument("--lora_alpha", default=16, type=int)
    p.add_argument("--lora_dropout", default=0.05, type=float)
    p.add_argument("--lora_target_modules", default="gate_proj,up_proj,down_proj", type=str)
    p.add_argument("--sft_modules", default="mask_decoder,text_hidden_fcs", type=str)      # train_ds_medplib.py:54 (+ wg, lm_head, embed_tokens)
    p.add_argument("--use_residual", type=bool, default=False)        # argparse bool as in train_ds_medplib.py:131
    p.add_argument("--router_aux_loss_coef", type=float, default=0.0)
    p.add_argument("--ep_size", type=int, default=1)
    add_layout_flags(p)
    p.add_argument("--seed", default=42, type=int)
    return p.parse_args(argv)


def add_layout_flags(p):
    """The ICL front-end and MoE layout flags of the reference (train_ds_medplib.py:73-76, 125-126, 134, 97); shared with infer.py."""
    p.add_argument("--mask_encoder_token_count", type=int, default=64)
    p.add_argument("--mm_token_compress", action="store_true", default=False)
    p.add_argument("--mm_compressed_token_count", type=int, default=256)
    p.add_argument("--moe_mode", type=str, default="dense", choices=["first_half", "second_half", "sparse", "dense"])
    p.add_argument("--moe_layers_idx", type=lambda s: [int(x) for x in s.split(",")] if s else None, default=None)
    p.add_argument("--out_dim", default=256, type=int)


class SyntheticDataset(torch.utils.data.Dataset):
    """SURVEY §8(d) synthetic samples already in batch form (one item = one collated micro-batch)."""

    def __init__(self, cfg, batch, n, seed, tiny):
        self.cfg, self.batch, self.n, self.seed, self.tiny = cfg, batch, n, seed, tiny

    def __len__(self):
        return self.n

    def __getitem__(self, i):
        return synth_batch(self.cfg, self.batch, self.seed + i, self.tiny)


def synth_batch(cfg, B, seed, tiny=False):
    """images ~ N(0,1), a 64-token prompt with one <image> placeholder bracketed by <im_start>/<im_end>, <SEG> near the end, labels
    on the tail, one binary disc mask per sample."""
    g = torch.Generator().manual_seed(seed)
    L, V = 64, cfg.vocab_size
    ids = torch.randint(3, min(V, cfg.seg_token_idx) - 1, (B, L), generator=g)
    ids[:, 0] = 1
    ids[:, 34], ids[:, 35], ids[:, 36] = V - 2, -200, V - 1
    ids[:, 61] = cfg.seg_token_idx
    ids[:, 63] = 2
    labels = ids.clone(); labels[:, :56] = -100
    H = W = 96 if tiny else 336
    yy, xx = torch.meshgrid(torch.arange(H), torch.arange(W), indexing="ij")
    masks = []
    for _ in range(B):
        cy, cx = (torch.rand(2, generator=g) * H).tolist()
        r = H / 16 + H / 4 * torch.rand(1, generator=g).item()
        masks.append((((yy - cy) ** 2 + (xx - cx) ** 2) < r * r).float())
    return {"images": torch.randn(B, 3, cfg.sam_image_size, cfg.sam_image_size, generator=g),
            "images_clip": torch.randn(B, 3, cfg.clip_image_size, cfg.clip_image_size, generator=g),
            "input_ids": ids, "labels": labels, "attention_mask": torch.ones(B, L, dtype=torch.bool), "masks_list": masks,
            "label_list": [torch.full((H, W), 255.0) for _ in range(B)], "resize_list": [(cfg.sam_image_size,) * 2] * B,
            "valid_mask_bool": [[True]] * B, "offset": None, "region_masks": [], "inference": False, "seg_flag": True}


def dict_to_device(batch, device):
    """utils.dict_to_cuda + the bf16 cast of the two image tensors (train_ds_medplib.py:583-591)."""
    out = {}
    for k, v in batch.items():
        if torch.is_tensor(v) and k in ("images", "images_clip"):
            v = v.to(device, non_blocking=True).to(torch.bfloat16)
        elif torch.is_tensor(v) and k not in ("input_ids", "labels", "attention_mask", "offset"):      # index tensors stay on the host
            v = v.to(device, non_blocking=True)
        elif isinstance(v, list) and v and torch.is_tensor(v[0]):
            v = [t.to(device, non_blocking=True) for t in v]
        out[k] = v
    return out


def build_model(args, device):
    kw = dict(moe_enable=args.moe_enable and not args.lisa, num_experts=args.num_experts, top_k_experts=args.top_k_experts,
              capacity_factor=args.capacity_factor, eval_capacity_factor=args.eval_capacity_factor, min_capacity=args.min_capacity,
              use_residual=getattr(args, "use_residual", False),
              router_aux_loss_coef=args.router_aux_loss_coef, ce_loss_weight=args.ce_loss_weight, dice_loss_weight=args.dice_loss_weight,
              bce_loss_weight=args.bce_loss_weight, iou_loss_weight=args.iou_loss_weight, focal_loss_weight=args.focal_loss_weight,
              train_mask_decoder=args.train_mask_decoder,
              icl_mask_encoder=bool(getattr(args, "icl_mask_encoder", False)), mask_encoder_token_count=getattr(args, "mask_encoder_token_count", 64),
              mm_token_compress=bool(getattr(args, "mm_token_compress", False)),
              mm_compressed_token_count=getattr(args, "mm_compressed_token_count", 256), out_dim=getattr(args, "out_dim", 256))
    n_layers = 2 if args.model_size == "tiny" else 32
    if kw["moe_enable"] and (getattr(args, "moe_layers_idx", None) is not None or getattr(args, "moe_mode", "dense") != "dense"):
        from .checkpoint import moe_layer_indices                       # medplib_moe_llama.py:575-596
        kw["moe_layers_idx"] = moe_layer_indices(n_layers, getattr(args, "moe_mode", "dense"), getattr(args, "moe_layers_idx", None))
    cfg = MedPLIBConfig.tiny(sam_depth=2, **kw) if args.model_size == "tiny" else MedPLIBConfig.medplib_7b(**kw)
    assert cfg.num_hidden_layers == n_layers
    model = (LISAForCausalLM if args.lisa else MedPLIBForCausalLM)(cfg, device=device)
    if args.version:
        model.load_hf_state_dict(torch.load(args.version, map_location="cpu"))
    if args.vision_pretrained:
        model.load_sam_state_dict(torch.load(args.vision_pretrained, map_location="cpu")["model"])
    if getattr(args, "lora_r", 0) > 0:
        # get_peft_model(LoraConfig(...)) (train_ds_medplib.py:262-303): adapters on the decoder's MLP projections, dense decoder only
        model.enable_lora(args.lora_r, args.lora_alpha, args.lora_dropout, args.lora_target_modules,
                          sft_modules=getattr(args, "sft_modules", "mask_decoder,text_hidden_fcs"))
    if args.ep_size > 1:
        from .expert_parallel import ExpertParallel, build_groups, build_host_group
        ep_group, _ = build_groups(args.ep_size)
        model.model.llm.enable_expert_parallel(ExpertParallel(ep_group, args.ep_size, cfg.num_experts, host_group=build_host_group(args.ep_size)))
    return cfg, model


def main(argv=None):
    args = parse_args(argv)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", str(args.local_rank)))
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group(backend="nccl")
    torch.manual_seed(args.seed)
    cfg, model = build_model(args, device)
    total_steps = args.epochs * args.steps_per_epoch
    ds_config = {"train_micro_batch_size_per_gpu": args.batch_size, "gradient_accumulation_steps": args.grad_accumulation_steps,
                 "optimizer": {"type": "AdamW", "params": {"lr": args.lr, "weight_decay": 0.0, "betas": (args.beta1, args.beta2)}},
                 "scheduler": {"type": "WarmupDecayLR", "params": {"total_num_steps": total_steps, "warmup_min_lr": 0, "warmup_max_lr": args.lr,
                                                                     "warmup_num_steps": max(1, args.steps_per_epoch // 100), "warmup_type": "linear"}},
                 "gradient_clipping": 1.0, "bf16": {"enabled": True}}                     # train_ds_medplib.py:383-420
    eng, optimizer, _, scheduler = E.initialize(model=model, model_parameters=model.trainable_parameters(args.sft_modules), config=ds_config)
    if args.dataset == "synthetic":
        data = SyntheticDataset(cfg, args.batch_size, 1 << 30, args.seed + 1000 * rank, args.model_size == "tiny")
        val = SyntheticDataset(cfg, 1, 4, args.seed + 7, args.model_size == "tiny")
    else:
        mod, _, fn = args.dataset.partition(":")
        data, val = getattr(importlib.import_module(mod), fn)(args, cfg)
    ckpt_dir = os.path.join(args.log_dir, "ckpt_model")
    resume = args.resume or (ckpt_dir if args.auto_resume and os.path.exists(os.path.join(ckpt_dir, "latest")) else "")
    start_epoch = 0
    if resume:
        eng.load_checkpoint(resume)                                                      # train_ds_medplib.py:453-470
        start_epoch = eng.global_steps // args.steps_per_epoch
        if rank == 0:
            print(f"resume training from {resume}, start from epoch {start_epoch} (global step {eng.global_steps})")
    if args.eval_only:
        return validate(val, eng, device, rank)
    it = eng.micro_steps                 # a resumed run continues with the micro-batches after the ones it already consumed
    history = []
    for epoch in range(start_epoch, args.epochs):
        eng.train()
        meter = E.AverageMeterPack(list(LOSS_KEYS), device)
        t_end = time.time()
        for local_step in range(eng.global_steps % args.steps_per_epoch, args.steps_per_epoch):
            for _ in range(args.grad_accumulation_steps):
                batch = dict_to_device(data[it], device); it += 1
                out = eng(**batch)
                meter.update_many({k: out[k].detach() for k in LOSS_KEYS}, n=batch["images"].shape[0])
                eng.backward(out["loss"])
                eng.step()
            if local_step % args.print_freq == 0:
                avg = meter.all_reduce()
                dt = time.time() - t_end
                t_end = time.time()
                history.append(avg["loss"])
                if rank == 0:
                    print(f"Epoch: [{epoch}][{local_step + 1}/{args.steps_per_epoch}] step {eng.global_steps} lr {eng.get_lr()[0]:.3e} "
                          f"time {dt:.3f} " + " ".join(f"{k} {avg[k]:.4f}" for k in ("loss", "ce_loss", "unscale_mask_loss")), flush=True)
                meter = E.AverageMeterPack(list(LOSS_KEYS), device)
            if local_step != 0 and local_step % args.save_steps == 0:
                eng.save_checkpoint(ckpt_dir)                                             # train_ds_medplib.py:693-698
        if not args.no_eval:
            validate(val, eng, device, rank)
    eng.save_checkpoint(ckpt_dir)
    return history


@torch.no_grad()
def validate(val, eng, device, rank):
    """validate() (train_ds_medplib.py:721-800): per sample inference forward, threshold 0.1, gIoU / cIoU / IoU / Dice meters."""
    eng.eval()
    meters = metrics.SegMeters()
    for i in range(len(val)):
        batch = dict_to_device(val[i], device)
        metrics.validate_batch(eng.module, batch, meters)
    meters.all_reduce(device)
    s = meters.summary()
    if rank == 0:
        print("giou: {giou:.4f}, ciou: {ciou:.4f}, iou: {iou:.4f}, dice: {dice:.4f}".format(**s), flush=True)
    eng.train()
    return s


if __name__ == "__main__":
    main()
