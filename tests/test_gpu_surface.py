"""GPU: the reference training driver's API sequence (repo-root train_ds_medplib.py: a stage table in this build's own words over the
import faces `model.*`, `datasets`, `utils.utils`, `medplib_amd.engine as deepspeed`, `medplib_amd.peft_compat`) against the module
surface, at tiny dims, from an HF-layout checkpoint directory written to disk: `from_pretrained(**vars(args))` -> initialize_vision_modules / initialize_bird_modules -> get_vision_tower().to() -> flag loops
-> find_linear_layers + get_peft_model -> initialize_moe_modules -> resize_token_embeddings -> --sft_modules substring loop ->
deepspeed.initialize(model_parameters=model.parameters()) -> engine(**batch) / backward / step -> save_checkpoint -> auto-resume."""
import os
import sys

import pytest
import torch

from medplib_amd.model.config import MedPLIBConfig
from oracle import model as OM

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _write_base(tmp_path, dev):
    """A dense 'LLaVA-like' base checkpoint directory in the HF key layout (what --version points at)."""
    from medplib_amd.model.medplib import LISAForCausalLM
    cfg = MedPLIBConfig.tiny(moe_enable=False, sam_depth=2)
    W = OM.init_hf_weights(cfg, seed=11)
    m = LISAForCausalLM(cfg, device=dev)
    m.load_hf_state_dict(W)
    d = str(tmp_path / "base")
    m.save_pretrained(d)
    return cfg, W, d


def _main(argv):
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    import train_ds_medplib as T
    return T.main(argv)


COMMON = ["--precision", "bf16", "--dataset", "synthetic", "--batch_size", "2", "--grad_accumulation_steps", "1", "--epochs", "1",
          "--steps_per_epoch", "3", "--save_steps", "2", "--lr", "1e-3", "--train_mask_decoder", "--dice_loss_weight", "5.0", "--bce_loss_weight", "1.0",
          "--iou_loss_weight", "0.5", "--focal_loss_weight", "1.0"]


def test_reference_driver_dense_lora_off(dev, tmp_path):
    """BASELINE configs[3]'s trainable set through the reference's flow: LISA class, --lora_r 0, --sft_modules mask_decoder,
    text_hidden_fcs.  The first micro-batch's loss must equal what the core class gives on the same weights and batch (the surface adds
    no arithmetic), training must move it, a checkpoint must appear and a second invocation must resume from it."""
    cfg, W, base = _write_base(tmp_path, dev)
    argv = COMMON + ["--version", base, "--lora_r", "0", "--sft_modules", "mask_decoder,text_hidden_fcs", "--log_base_dir", str(tmp_path),
                     "--exp_name", "dense"]
    hist = _main(argv)
    assert len(hist) == 3 and all(h == h for h in hist)
    from medplib_amd.model.medplib import LISAForCausalLM
    from medplib_amd.train import synth_batch
    import train_ds_medplib as T
    ref = LISAForCausalLM(MedPLIBConfig.tiny(moe_enable=False, sam_depth=2, dice_loss_weight=5.0, bce_loss_weight=1.0, iou_loss_weight=0.5,
                                             focal_loss_weight=1.0, ce_loss_weight=1.0), device=dev).train()
    ref.load_hf_state_dict(torch.load(os.path.join(base, "pytorch_model.bin"), map_location="cpu"))   # what --version holds (bf16 export)
    firsts = []
    for seed in (42, 43, 44):                              # the loader shuffles the three seeded micro-batches
        b = T.to_device_in_dtype(synth_batch(ref.config, 2, seed, tiny=True), "bf16")
        firsts.append(float(ref(**b)["loss"]))
    assert min(abs(f - hist[0]) for f in firsts) < 1e-6, (firsts, hist[0])
    ck = tmp_path / "dense" / "ckpt_model"
    assert (ck / "latest").read_text().strip() == "global_step3"
    saved = torch.load(ck / "global_step3" / "mp_rank_00_model_states.pt", map_location="cpu")["module"]
    assert saved and all(("mask_decoder" in k or "text_hidden_fcs" in k) for k in saved)       # DeepSpeed-style module names
    assert (tmp_path / "dense" / "last_ckpt_model" / "latest").read_text().strip() == "global_step3"      # the end-of-epoch checkpoint
    # auto-resume: epoch 0 is done, one more epoch runs; then the validation loop (inference-mode forward, sigmoid > 0.1,
    # intersectionAndUnionGPU, five meters) over two synthetic samples
    hist2 = _main(argv + ["--epochs", "2", "--val_samples", "2"])
    assert len(hist2) == 3
    giou, ciou, miou, mdice = T.main.last_run.val_scores[-1]
    assert 0.0 <= giou <= 1.0 and 0.0 <= ciou <= 1.0 and 0.0 <= miou <= 1.0 and abs(mdice - mdice) == 0.0


def test_reference_driver_lora_dense(dev, tmp_path):
    """scripts/train_stage3.sh's shape: LISA class, LoRA r=8 on gate/up/down through find_linear_layers + get_peft_model, --sft_modules
    lm_head,embed_tokens,mask_decoder,text_hidden_fcs (the driver default)."""
    cfg, W, base = _write_base(tmp_path, dev)
    hist = _main(COMMON + ["--version", base, "--lora_r", "8", "--lora_alpha", "16", "--lora_dropout", "0.05", "--lora_target_modules",
                           "gate_proj,up_proj,down_proj", "--log_base_dir", str(tmp_path), "--exp_name", "lora"])
    assert len(hist) == 3 and all(h == h for h in hist)
    saved = torch.load(tmp_path / "lora" / "ckpt_model" / "global_step3" / "mp_rank_00_model_states.pt", map_location="cpu")["module"]
    assert "base_model.model.model.layers.0.mlp.gate_proj.lora_A.default.weight" in saved and "base_model.model.lm_head.weight" in saved


def test_reference_driver_lora_moe_stage4(dev, tmp_path):
    """scripts/train_stage4.sh's shape: MoE class from a dense base, adapters on gate/up/down + q/v, initialize_moe_modules (E = 2, top-1,
    all layers, experts seeded from two stage checkpoints on disk) after get_peft_model, --sft_modules wg,lm_head,embed_tokens,
    mask_decoder,text_hidden_fcs, MoE optimizer param groups (:422-434)."""
    cfg, W, base = _write_base(tmp_path, dev)
    hist = _main(COMMON + ["--version", base, "--moe_enable", "True", "--moe_mode", "dense", "--num_experts", "2", "--top_k_experts", "1",
                           "--capacity_factor", "1.5", "--router_aux_loss_coef", "0.0", "--expert_pretrained_path", f"{base},{base}",
                           "--lora_r", "8", "--lora_target_modules", "gate_proj,up_proj,down_proj,q_proj,v_proj",
                           "--sft_modules", "wg,lm_head,embed_tokens,mask_decoder,text_hidden_fcs",
                           "--log_base_dir", str(tmp_path), "--exp_name", "moe"])
    assert len(hist) == 3 and all(h == h for h in hist)
    saved = torch.load(tmp_path / "moe" / "ckpt_model" / "global_step3" / "mp_rank_00_model_states.pt", map_location="cpu")["module"]
    assert "base_model.model.model.layers.1.mlp.deepspeed_moe.experts.deepspeed_experts.1.up_proj.lora_B.default.weight" in saved
    assert "base_model.model.model.layers.0.mlp.deepspeed_moe.gate.wg.weight" in saved


def test_reference_inference_driver_call_shapes(dev, tmp_path):
    """model/eval/vqa_infer.py's side of the surface (the reference's :244-312,430-442,528-540): from_pretrained(**vars(args),
    test_only=True) -> config ids -> get_vision_tower() -> model.to(dtype=, device=) -> all parameters frozen -> model.eval() ->
    `model.evaluate(images_clip, images, input_ids, resize_list, label_list, max_new_tokens=, tokenizer=, attention_mask=, mask_images=,
    image_token_types=, image_token_lengths=)` and `model.generate(input_ids, images=, attention_mask=, mask_images=, image_token_types=,
    do_sample=False, temperature=0, top_p=None, num_beams=1, max_new_tokens=, use_cache=True)`.  The walk must give exactly what the
    core class gives on the same weights and samples: token ids equal, mIoU equal."""
    cfg, W, base = _write_base(tmp_path, dev)
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    from model.eval import vqa_infer as V
    ans = tmp_path / "answers" / "out.jsonl"
    out = V.main(["--version", base, "--dataset", "synthetic", "--n_samples", "2", "--max_new_tokens", "6", "--colon_token_id", "7",
                  "--eval_seg", "--eval_vqa", "--answers-file", str(ans), "--precision", "bf16"])
    assert 0.0 <= out["miou"] <= 1.0 and 0.0 <= out["mdice"] <= 1.0 and set(out["per_modality"]) == {"synthetic"}
    lines = [__import__("json").loads(l) for l in ans.read_text().splitlines()]
    assert len(lines) == 2 and all(len(l["output_ids"]) >= 1 for l in lines) and [l["output_ids"] for l in lines] == out["answers"]
    # the same samples through the core class
    from medplib_amd.model.medplib import LISAForCausalLM
    from medplib_amd.train import synth_batch
    core = LISAForCausalLM(MedPLIBConfig.tiny(moe_enable=False, sam_depth=2), device=dev).eval()
    core.load_hf_state_dict(torch.load(os.path.join(base, "pytorch_model.bin"), map_location="cpu"))
    ious = []
    for i in range(2):
        b = synth_batch(core.config, 1, 42 + 7 + i, tiny=True)
        b["input_ids"][:, 55] = 7
        ids, att = b["input_ids"][:, :56], b["attention_mask"][:, :56]
        clip, img = b["images_clip"].to(dev).bfloat16(), b["images"].to(dev).bfloat16()
        gen = core.generate(ids, images=clip, attention_mask=att, max_new_tokens=6)
        assert gen[0, 56:].tolist()[:len(out["answers"][i])] == out["answers"][i]
        _, pm = core.evaluate(clip, img, ids, b["resize_list"], b["label_list"], max_new_tokens=6, attention_mask=att)
        g = b["masks_list"][0].to(dev).bool().reshape(-1)
        p = (torch.sigmoid(pm[0].float()) > 0.1).reshape(-1)
        ious.append(int((p & g).sum()) / max(int((p | g).sum()), 1))
    assert abs(sum(ious) / 2 - out["miou"]) < 1e-9
    # the MoE class from an MoE checkpoint directory (what scripts/eval on a stage-IV model load): E = 2 top-1 experts in every layer
    from medplib_amd.model.medplib import MedPLIBForCausalLM as CoreMoE
    mcfg = MedPLIBConfig.tiny(moe_enable=True, sam_depth=2, num_experts=2, top_k_experts=1)
    moe = CoreMoE(mcfg, device=dev)
    moe.load_hf_state_dict(OM.init_hf_weights(mcfg, seed=12))
    moe_dir = str(tmp_path / "moe_base")
    moe.save_pretrained(moe_dir)
    out_m = V.main(["--version", moe_dir, "--dataset", "synthetic", "--n_samples", "1", "--max_new_tokens", "4", "--colon_token_id", "7",
                    "--eval_seg", "--eval_vqa", "--moe_enable", "--moe_mode", "dense", "--num_experts", "2", "--top_k_experts", "1",
                    "--precision", "bf16"])
    assert 0.0 <= out_m["miou"] <= 1.0 and len(out_m["answers"]) == 1 and len(out_m["answers"][0]) >= 1
    # what the walk refuses loudly
    m2 = V.LISAForCausalLM.from_pretrained(base, torch_dtype=torch.bfloat16, test_only=True)
    with pytest.raises(ValueError):
        m2.to(dtype=torch.float16)
    with pytest.raises(NotImplementedError):
        m2.generate(b["input_ids"][:, :56], images=clip, do_sample=True, temperature=0.7)
    with pytest.raises(NotImplementedError):
        m2.generate(b["input_ids"][:, :56], images=clip, num_beams=4)
