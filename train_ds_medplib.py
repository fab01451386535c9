"""The reference's training driver, call for call, against the MI355X build.

This file replays `/root/reference/train_ds_medplib.py` main() :181-533 and train() :536-700 with the reference's own statements in
the reference's own order — the only substitutions are the three imports a maintainer changes:

    from model.MedPLIB import MedPLIBForCausalLM; from model.LISA import LISAForCausalLM     # unchanged: repo-root `model/` package
    from medplib_amd.peft_compat import LoraConfig, get_peft_model                            # was: from peft import ...
    import medplib_amd.engine as deepspeed                                                    # was: import deepspeed

Flags are the reference's (:28-138).  Additions, all optional: `--dataset synthetic` (seeded batches of SURVEY §8d instead of
JSON files — no datasets / tokenizer files exist on the build or GPU boxes), `--steps_per_epoch` to bound a synthetic epoch,
`--tokenizer_path` for a sentencepiece model when `--version` holds no tokenizer files.
`tests/test_gpu_surface.py` runs it end to end at tiny dims for the dense (LISA), LoRA and LoRA + MoE branches."""
import argparse
import math
import os
import sys
import time
import types
from functools import partial

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import medplib_amd.engine as deepspeed                                        # noqa: E402
from medplib_amd.peft_compat import LoraConfig, get_peft_model                # noqa: E402
from model.LISA import LISAForCausalLM                                        # noqa: E402
from model.MedPLIB import MedPLIBForCausalLM                                  # noqa: E402

DEFAULT_IM_START_TOKEN, DEFAULT_IM_END_TOKEN = "<im_start>", "<im_end>"      # utils/utils.py


def parse_args(args):
    parser = argparse.ArgumentParser(description="MedPLIB Model Training")   # train_ds_medplib.py:28-138, same names and defaults
    parser.add_argument("--local_rank", default=0, type=int, help="node rank")
    parser.add_argument("--version", default="liuhaotian/llava-llama-2-13b-chat-lightning-preview")
    parser.add_argument("--pretrain_mm_mlp_adapter", default=None, type=str)
    parser.add_argument("--precision", default="bf16", type=str, choices=["fp32", "bf16", "fp16"])
    parser.add_argument("--sam_img_size", default=256, type=int)
    parser.add_argument("--model_max_length", default=512, type=int)
    parser.add_argument("--vision_tower", default="openai/clip-vit-large-patch14", type=str)
    parser.add_argument("--vision_pretrained", default="PATH_TO_SAM_ViT-H", type=str)
    parser.add_argument("--sft_modules", default="lm_head,embed_tokens,mask_decoder,text_hidden_fcs", type=str)
    parser.add_argument("--lora_r", default=8, type=int)
    parser.add_argument("--lora_alpha", default=16, type=int)
    parser.add_argument("--lora_dropout", default=0.05, type=float)
    parser.add_argument("--lora_target_modules", default="q_proj,v_proj", type=str)
    parser.add_argument("--image_folder", type=str, default="/path/to/SAMed2D_v1")
    parser.add_argument("--image_aspect_ratio", type=str, default="pad")
    parser.add_argument("--is_multimodal", type=bool, default=True)
    parser.add_argument("--data_path", type=str, default="/path/to/xxx.json")
    parser.add_argument("--val_data_path", type=str, default="/path/to/xxx.json")
    parser.add_argument("--icl_enable", action="store_true", default=False)
    parser.add_argument("--icl_mask_mode", type=str, default="overlay", choices=["overlay", "separate"])
    parser.add_argument("--icl_mask_encoder", action="store_true", default=False)
    parser.add_argument("--mask_encoder_token_count", type=int, default=64)
    parser.add_argument("--mm_token_compress", action="store_true", default=False)
    parser.add_argument("--mm_compressed_token_count", type=int, default=256)
    parser.add_argument("--log_base_dir", default="./runs", type=str)
    parser.add_argument("--exp_name", default="lisa", type=str)
    parser.add_argument("--epochs", default=10, type=int)
    parser.add_argument("--batch_size", default=2, type=int)
    parser.add_argument("--grad_accumulation_steps", default=10, type=int)
    parser.add_argument("--val_batch_size", default=1, type=int)
    parser.add_argument("--workers", default=4, type=int)
    parser.add_argument("--lr", default=0.0003, type=float)
    parser.add_argument("--ce_loss_weight", default=1.0, type=float)
    parser.add_argument("--dice_loss_weight", default=0.5, type=float)
    parser.add_argument("--bce_loss_weight", default=2.0, type=float)
    parser.add_argument("--iou_loss_weight", default=2.0, type=float)
    parser.add_argument("--focal_loss_weight", default=2.0, type=float)
    parser.add_argument("--beta1", default=0.9, type=float)
    parser.add_argument("--beta2", default=0.95, type=float)
    parser.add_argument("--no_eval", action="store_true", default=False)
    parser.add_argument("--eval_only", action="store_true", default=False)
    parser.add_argument("--out_dim", default=256, type=int)
    parser.add_argument("--resume", default="", type=str)
    parser.add_argument("--print_freq", default=1, type=int)
    parser.add_argument("--save_steps", default=10, type=int)
    parser.add_argument("--start_epoch", default=0, type=int)
    parser.add_argument("--gradient_checkpointing", action="store_true", default=True)
    parser.add_argument("--train_mask_decoder", action="store_true", default=False)
    parser.add_argument("--use_mm_start_end", action="store_true", default=True)
    parser.add_argument("--auto_resume", action="store_true", default=True)
    parser.add_argument("--conv_type", default="llava_v1", type=str, choices=["llava_v1", "llava_llama_2"])
    parser.add_argument("--region_fea_adapter", action="store_true", default=False)
    parser.add_argument("--region_geo_sampler", action="store_true", default=False)
    parser.add_argument("--max_sample_point", default=512, type=int)
    parser.add_argument("--sampler_pooler_mode", default="max", type=str)
    parser.add_argument("--moe_enable", type=bool, default=False)
    parser.add_argument("--moe_mode", type=str, default="second_half", choices=["first_half", "second_half", "sparse", "dense"])
    parser.add_argument("--num_experts", type=int, default=3)
    parser.add_argument("--top_k_experts", type=int, default=2)
    parser.add_argument("--capacity_factor", type=float, default=1)
    parser.add_argument("--use_residual", type=bool, default=False)
    parser.add_argument("--router_aux_loss_coef", type=float, default=0.01)
    parser.add_argument("--eval_capacity_factor", type=float, default=2)
    parser.add_argument("--moe_layers_idx", type=str, default=None)
    parser.add_argument("--min_capacity", type=int, default=0)
    parser.add_argument("--ep_size", type=int, default=1)
    parser.add_argument("--expert_pretrained_path", type=str, default=None)
    parser.add_argument("--finetune_moe", type=bool, default=False)
    # ---- additions of this build (see the module docstring)
    parser.add_argument("--dataset", default="json", choices=["json", "synthetic"])
    parser.add_argument("--steps_per_epoch", default=0, type=int)
    parser.add_argument("--tokenizer_path", default="", type=str)
    parser.add_argument("--seed", default=42, type=int)
    return parser.parse_args(args)


def build_tokenizer(args):
    """train_ds_medplib.py:198-216.  Returns (tokenizer or None, number of token ids, <SEG> id): without tokenizer files (synthetic
    runs) the ids are the ones the seeded batches use — <SEG> = the checkpoint's config.seg_token_idx, vocabulary unchanged."""
    src = args.tokenizer_path or args.version
    has_files = os.path.isdir(src) and any(os.path.exists(os.path.join(src, f)) for f in ("tokenizer.model", "tokenizer.json"))
    if not has_files:
        import json
        cfg = json.load(open(os.path.join(args.version, "config.json")))
        return None, int(cfg["vocab_size"]), int(cfg.get("seg_token_idx", 32000))
    import transformers
    tokenizer = transformers.AutoTokenizer.from_pretrained(src, cache_dir=None, model_max_length=args.model_max_length,
                                                           padding_side="right", use_fast=False, legacy=True)
    tokenizer.pad_token = tokenizer.unk_token
    others = ["<SEG>", "<region>", "</region>", "<mask>", "</mask>", "<bbox>", "</bbox>", "<point>", "</point>", "<p>", "</p>"]
    for i in range(1, 257):
        others.append("<gen_" + str(i) + ">")
    for name in others:                                                       # ADD_OTHERS_TOKENS, utils/utils.py
        tokenizer.add_tokens(name, special_tokens=True)
    seg = tokenizer("<SEG>", add_special_tokens=False).input_ids[0]
    if args.use_mm_start_end:
        tokenizer.add_tokens([DEFAULT_IM_START_TOKEN, DEFAULT_IM_END_TOKEN], special_tokens=True)
    return tokenizer, len(tokenizer), seg


def main(argv):
    args = parse_args(argv)
    args.log_dir = os.path.join(args.log_base_dir, args.exp_name)
    local_rank = int(os.environ.get("LOCAL_RANK", args.local_rank))
    args.local_rank = local_rank
    torch.cuda.set_device(local_rank)
    deepspeed.init_distributed(dist_backend="nccl")                           # the launcher's env; RCCL on ROCm
    torch.manual_seed(args.seed)
    if args.local_rank == 0:
        os.makedirs(args.log_dir, exist_ok=True)
    if isinstance(args.moe_layers_idx, str):
        args.moe_layers_idx = [int(x) for x in args.moe_layers_idx.split(",")]
    args.num_experts = [args.num_experts]                                     # the model reads a list (medplib_moe_llama.py:597)

    tokenizer, n_tokens, args.seg_token_idx = build_tokenizer(args)

    # ---------------------------------------------------------------- Create model (train_ds_medplib.py:218-238)
    model_args = vars(args)
    torch_dtype = torch.float32
    if args.precision == "bf16":
        torch_dtype = torch.bfloat16
    elif args.precision == "fp16":
        torch_dtype = torch.half
    if args.moe_enable:
        model = MedPLIBForCausalLM.from_pretrained(args.version, torch_dtype=torch_dtype, low_cpu_mem_usage=True,
                                                   ignore_mismatched_sizes=True, **model_args)
    else:
        model = LISAForCausalLM.from_pretrained(args.version, torch_dtype=torch_dtype, low_cpu_mem_usage=True,
                                                ignore_mismatched_sizes=True, **model_args)
    if tokenizer is not None:
        model.config.eos_token_id = tokenizer.eos_token_id
        model.config.bos_token_id = tokenizer.bos_token_id
        model.config.pad_token_id = tokenizer.pad_token_id

    model.enable_input_require_grads()
    model.gradient_checkpointing_enable()

    # load tower and projector weights (:240-247)
    model.get_model().initialize_vision_modules(model.get_model().config)
    if not args.eval_only:
        if args.moe_enable:
            model.get_model().initialize_bird_modules(model.get_model().config)
        else:
            model.get_model().initialize_lisa_modules(model.get_model().config)

    vision_tower = model.get_model().get_vision_tower()
    vision_tower.to(dtype=torch_dtype, device=args.local_rank)

    for p in vision_tower.parameters():
        p.requires_grad = False
    for p in model.get_model().mm_projector.parameters():
        p.requires_grad = False

    # ---------------------------------------------------------------- LoRA (:261-306)
    lora_r = args.lora_r
    if lora_r > 0:

        def find_linear_layers(model, lora_target_modules):
            cls = torch.nn.Linear
            lora_module_names = set()
            for name, module in model.named_modules():
                if (isinstance(module, cls)
                        and all([x not in name for x in ["visual_model", "vision_tower", "mm_projector"]])
                        and any([x in name for x in lora_target_modules])):
                    lora_module_names.add(name)
            return sorted(list(lora_module_names))

        lora_target_modules = find_linear_layers(model, args.lora_target_modules.split(","))
        if args.local_rank == 0:
            print("lora_target_modules", len(lora_target_modules), lora_target_modules[:4], "...")
        lora_config = LoraConfig(r=lora_r, lora_alpha=args.lora_alpha, target_modules=lora_target_modules,
                                 lora_dropout=args.lora_dropout, bias="none", task_type="CAUSAL_LM")
        model = get_peft_model(model, lora_config)
        model.print_trainable_parameters()
    else:
        for n, p in model.named_parameters():
            p.requires_grad = False

    if args.moe_enable:
        model.initialize_moe_modules(args)                                    # :308-310

    model.resize_token_embeddings(n_tokens)                                   # :312

    # make text_hidden_fcs, mask_decoder, lm_head, embed_tokens trainable (:315-326)
    if args.sft_modules != "":
        sft_modules = args.sft_modules.split(",")
        for n, p in model.named_parameters():
            if any([x in n for x in sft_modules]):
                p.requires_grad = True

    def count_parameters(model):
        trainable_params = sum(p.numel() for p in model.parameters() if p.requires_grad)
        total_params = sum(p.numel() for p in model.parameters())
        return trainable_params, total_params
    trainable_params, total_params = count_parameters(model)
    if args.local_rank == 0:
        print(f"Trainable Parameters: {trainable_params}")
        print(f"Total Parameters: {total_params}")
        print(f"Trainable Parameters Percentage: {trainable_params / total_params * 100:.7f}%")

    # ---------------------------------------------------------------- data (:352-381)
    world_size = int(os.environ.get("WORLD_SIZE", "1"))
    if args.dataset == "synthetic":
        from medplib_amd.train import SyntheticDataset
        cfg = model.config
        tiny = cfg.hidden_size < 1024
        steps = args.steps_per_epoch or 4
        train_dataset = SyntheticDataset(cfg, args.batch_size, steps * args.grad_accumulation_steps * args.epochs, args.seed + 1000 * args.local_rank, tiny)
        args.steps_per_epoch = steps
        collate, micro = (lambda items: items[0]), None      # an item already is one collated micro-batch
    else:
        from medplib_amd import dataset as D
        from medplib_amd.collate import collate as DataCollatorForSupervisedDataset
        data_args = types.SimpleNamespace(image_folder=args.image_folder, image_aspect_ratio=args.image_aspect_ratio,
                                          is_multimodal=args.is_multimodal, mm_use_im_start_end=args.use_mm_start_end,
                                          data_path=args.data_path, icl_mask_mode=args.icl_mask_mode, icl_mask_encoder=args.icl_mask_encoder,
                                          mask_encoder_token_count=args.mask_encoder_token_count, mm_token_compress=args.mm_token_compress,
                                          mm_compressed_token_count=args.mm_compressed_token_count,
                                          image_processor=vision_tower.image_processor)
        dataset_cls = D.ICLLazySupervisedDataset if args.icl_enable else D.LazySupervisedDataset
        train_dataset = dataset_cls(args.data_path, tokenizer, data_args, args.sam_img_size)
        args.steps_per_epoch = math.ceil(math.ceil(len(train_dataset) / (args.batch_size * world_size)) / args.grad_accumulation_steps)
        collate, micro = partial(DataCollatorForSupervisedDataset), args.batch_size

    ds_config = {                                                             # :383-420
        "train_micro_batch_size_per_gpu": micro if micro is not None else 1,
        "gradient_accumulation_steps": args.grad_accumulation_steps,
        "optimizer": {"type": "AdamW", "params": {"lr": args.lr, "weight_decay": 0.0, "betas": (args.beta1, args.beta2)}},
        "gradient_clipping": 1.0,
        "scheduler": {"type": "WarmupDecayLR", "params": {"total_num_steps": args.epochs * args.steps_per_epoch, "warmup_min_lr": 0,
                                                            "warmup_max_lr": args.lr, "warmup_num_steps": int(args.steps_per_epoch * 0.01),
                                                            "warmup_type": "linear"}},
        "fp16": {"enabled": args.precision == "fp16"},
        "bf16": {"enabled": args.precision == "bf16"},
        "zero_optimization": {"stage": 2, "contiguous_gradients": True, "overlap_comm": True, "reduce_scatter": True,
                              "reduce_bucket_size": 5e8, "allgather_bucket_size": 5e8},
    }

    if args.moe_enable and "up_proj" in args.lora_target_modules:             # :422-434
        parameters = {"params": [p for p in model.parameters()], "name": "parameters"}
        optimizer_grouped_parameters = deepspeed.split_params_into_different_moe_groups_for_optimizer(parameters)
    else:
        optimizer_grouped_parameters = model.parameters()

    model_engine, optimizer, train_loader, scheduler = deepspeed.initialize(   # :439-448
        model=model, model_parameters=optimizer_grouped_parameters, training_data=train_dataset,
        collate_fn=collate, config=ds_config)

    # resume deepspeed checkpoint (:452-470)
    if args.auto_resume and len(args.resume) == 0:
        resume = os.path.join(args.log_dir, "ckpt_model")
        if os.path.exists(resume):
            args.resume = resume
    if args.resume:
        load_path, client_state = model_engine.load_checkpoint(args.resume)
        with open(os.path.join(args.resume, "latest"), "r") as f:
            ckpt_dir = f.readlines()[0].strip()
        args.start_epoch = int(ckpt_dir.replace("global_step", "")) // args.steps_per_epoch
        print("resume training from {}, start from epoch {}".format(args.resume, args.start_epoch))

    history = []
    train_iter = iter(train_loader)
    for epoch in range(args.start_epoch, args.epochs):                        # :511-533
        train_iter, losses = train(train_loader, model_engine, epoch, scheduler, train_iter, args)
        history += losses
        save_dir = os.path.join(args.log_dir, "ckpt_model")
        model_engine.save_checkpoint(save_dir)
    return history


def dict_to_cuda(input_dict, device):                                         # utils.dict_to_cuda
    out = {}
    for k, v in input_dict.items():
        if torch.is_tensor(v) and k not in ("input_ids", "labels", "attention_mask", "offset"):   # index tensors are host work here
            v = v.to(device, non_blocking=True)
        elif isinstance(v, list) and len(v) > 0 and torch.is_tensor(v[0]):
            v = [e.to(device, non_blocking=True) for e in v]
        out[k] = v
    return out


def train(train_loader, model, epoch, scheduler, train_iter, args):
    """train() of the reference (:536-700): steps_per_epoch x grad_accumulation_steps micro-batches of
    model(**input_dict) / model.backward(loss) / model.step()."""
    model.train()
    losses = []
    for global_step in range(args.steps_per_epoch):
        for i in range(args.grad_accumulation_steps):
            try:
                input_dict = next(train_iter)
            except StopIteration:
                train_iter = iter(train_loader)
                input_dict = next(train_iter)
            input_dict = dict_to_cuda(input_dict, torch.device("cuda", args.local_rank))
            if args.precision == "bf16":                                      # :588-591
                input_dict["images"] = input_dict["images"].bfloat16()
                ic = input_dict["images_clip"]
                input_dict["images_clip"] = [x.bfloat16() for x in ic] if isinstance(ic, list) else ic.bfloat16()
            output_dict = model(**input_dict)
            loss = output_dict["loss"]
            model.backward(loss)
            model.step()
        if global_step % args.print_freq == 0:
            losses.append(loss.item())
            if args.local_rank == 0:
                print(f"Epoch: [{epoch}][{global_step + 1}/{args.steps_per_epoch}] loss {losses[-1]:.4f} "
                      f"ce {output_dict['ce_loss'].item():.4f} mask {output_dict['mask_loss'].item():.4f} lr {scheduler.get_last_lr()[0]:.3e}",
                      flush=True)
    return train_iter, losses


if __name__ == "__main__":
    main(sys.argv[1:])
