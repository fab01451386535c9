"""CPU: the reference's module surface (medplib_amd/surface.py, model/MedPLIB.py, model/LISA.py, medplib_amd/peft_compat.py) — names,
classes, flags and the training plan they resolve to — exercised with the statements of the reference driver
(train_ds_medplib.py:250-326).  No kernels run here; the GPU twin (tests/test_gpu_surface.py) runs the whole driver."""
import types

import pytest
import torch

from medplib_amd.model.config import MedPLIBConfig
from medplib_amd.peft_compat import LoraConfig, get_peft_model
from oracle import model as OM


def _find_linear_layers(model, lora_target_modules):                 # train_ds_medplib.py:265-285, verbatim logic
    names = set()
    for name, module in model.named_modules():
        if (isinstance(module, torch.nn.Linear) and all(x not in name for x in ["visual_model", "vision_tower", "mm_projector"])
                and any(x in name for x in lora_target_modules)):
            names.add(name)
    return sorted(names)


def _driver_freeze(model):
    vt = model.get_model().get_vision_tower()
    vt.to(dtype=torch.bfloat16, device=0)
    for p in vt.parameters():
        p.requires_grad = False
    for p in model.get_model().mm_projector.parameters():
        p.requires_grad = False


def test_surface_names_are_the_hf_checkpoint_names():
    from model.LISA import LISAForCausalLM
    cfg = MedPLIBConfig.tiny(moe_enable=False, sam_depth=2)
    m = LISAForCausalLM(cfg, device="cpu")
    names = dict(m.named_parameters())
    W = OM.init_hf_weights(cfg)
    buffers = {"model.visual_model.prompt_encoder.pe_layer.positional_encoding_gaussian_matrix"}      # a buffer in the reference too
    assert set(W) - set(names) == buffers and not (set(names) - set(W))
    assert all(tuple(names[k].shape) == tuple(W[k].shape) for k in names)
    sd = m.state_dict()
    assert set(sd) >= set(names) and all(tuple(sd[k].shape) == tuple(names[k].shape) for k in names)
    # values go in and out through the same names
    m.load_state_dict(W, strict=False)
    sd = m.state_dict()
    for k in ("model.layers.1.mlp.gate_proj.weight", "model.layers.0.self_attn.k_proj.weight", "lm_head.weight",
              "model.text_hidden_fcs.0.2.bias", "model.mm_projector.2.weight"):
        assert torch.equal(sd[k].float().cpu(), W[k].to(sd[k].dtype).float()), k
    with pytest.raises(RuntimeError):
        m.load_state_dict({"lm_head.weight": torch.zeros(7, cfg.hidden_size)}, strict=False)
    r = m.load_state_dict({"lm_head.weight": torch.zeros(7, cfg.hidden_size)}, strict=False, ignore_mismatched_sizes=True)
    assert r.mismatched_keys == ["lm_head.weight"]


def test_driver_sequence_lora_off_trains_the_mask_tail_only():
    """Stage-III "LoRA off" (BASELINE configs[3]): lora_r = 0 -> every flag False, then --sft_modules mask_decoder,text_hidden_fcs."""
    from model.LISA import LISAForCausalLM
    cfg = MedPLIBConfig.tiny(moe_enable=False, sam_depth=2)
    model = LISAForCausalLM(cfg, device="cpu", train_mask_decoder=True)
    model.enable_input_require_grads(); model.gradient_checkpointing_enable()
    model.get_model().initialize_vision_modules(model.get_model().config)
    model.get_model().initialize_lisa_modules(model.get_model().config)
    _driver_freeze(model)
    for n, p in model.named_parameters():
        p.requires_grad = False
    model.resize_token_embeddings(cfg.vocab_size + 5)
    assert dict(model.named_parameters())["lm_head.weight"].shape[0] == cfg.vocab_size
    for n, p in model.named_parameters():
        if any(x in n for x in ["mask_decoder", "text_hidden_fcs"]):
            p.requires_grad = True
    on = sorted(n for n, p in model.named_parameters() if p.requires_grad)
    assert on and all(("mask_decoder" in n or "text_hidden_fcs" in n) for n in on)
    params = model.resolve_training_plan(model.parameters())
    assert sum(p.numel() for p in params) == sum(p.numel() for n, p in model.named_parameters() if p.requires_grad)
    assert all(not p.is_meta for p in params) and getattr(model.model, "lora", None) is None


def test_driver_sequence_lora_moe_stage4():
    """scripts/train_stage4.sh: MoE class, LoRA r=8 on gate/up/down + q/v, initialize_moe_modules AFTER get_peft_model (the wrapped MLP is
    deep-copied into the experts), --sft_modules wg,lm_head,embed_tokens,mask_decoder,text_hidden_fcs."""
    from model.MedPLIB import MedPLIBForCausalLM
    cfg = MedPLIBConfig.tiny(moe_enable=False, sam_depth=2)
    model = MedPLIBForCausalLM(cfg, device="cpu", train_mask_decoder=True)
    assert not model.model.llm.moe_layers                     # a dense base until initialize_moe_modules
    model.get_model().initialize_vision_modules(model.get_model().config)
    model.get_model().initialize_bird_modules(model.get_model().config)
    _driver_freeze(model)
    targets = _find_linear_layers(model, "gate_proj,up_proj,down_proj,q_proj,v_proj".split(","))
    assert len(targets) == 5 * cfg.num_hidden_layers and "model.layers.0.self_attn.q_proj" in targets
    model = get_peft_model(model, LoraConfig(r=8, lora_alpha=16, target_modules=targets, lora_dropout=0.05, bias="none", task_type="CAUSAL_LM"))
    names = [n for n, _ in model.named_parameters()]
    assert "base_model.model.model.layers.0.mlp.gate_proj.base_layer.weight" in names
    assert "base_model.model.model.layers.1.self_attn.v_proj.lora_B.default.weight" in names
    assert "base_model.model.model.layers.0.self_attn.k_proj.weight" in names            # not a target: unwrapped
    assert sorted(n for n, p in model.named_parameters() if p.requires_grad) == sorted(n for n in names if ".lora_" in n)
    args = types.SimpleNamespace(num_experts=[2], moe_mode="dense", moe_layers_idx=None, ep_size=1, top_k_experts=1, capacity_factor=1.5,
                                 eval_capacity_factor=2.0, min_capacity=0, use_residual=False, router_aux_loss_coef=0.0,
                                 expert_pretrained_path=None, moe_enable=True)
    model.initialize_moe_modules(args)
    model.resize_token_embeddings(cfg.vocab_size)
    named = dict(model.named_parameters())
    e1 = "base_model.model.model.layers.1.mlp.deepspeed_moe.experts.deepspeed_experts.1."
    assert named[e1 + "up_proj.lora_A.default.weight"].requires_grad and not named[e1 + "up_proj.base_layer.weight"].requires_grad
    assert named["base_model.model.model.layers.0.mlp.deepspeed_moe.gate.wg.weight"].dtype == torch.float32
    assert not any(".mlp.gate_proj." in n for n in named)     # the dense MLP names are gone
    for n, p in model.named_parameters():
        if any(x in n for x in "wg,lm_head,embed_tokens,mask_decoder,text_hidden_fcs".split(",")):
            p.requires_grad = True
    params = model.get_base_model().resolve_training_plan(model.parameters())
    lo = model.get_base_model().model.lora
    assert lo is not None and lo.targets == ("q_proj", "v_proj", "gate_proj", "up_proj", "down_proj") and lo.train_gate
    assert lo.full_param("lm_head.weight") is not None and lo.full_param("model.embed_tokens.weight") is not None
    # after resolution the handles of everything trainable are REAL parameters, under the reference's (peft) names
    on = {n: p for n, p in model.named_parameters() if p.requires_grad}
    assert all(not p.is_meta for p in on.values())
    assert e1 + "down_proj.lora_B.default.weight" in on and "base_model.model.lm_head.weight" in on
    assert {id(p) for p in params} == {id(p) for p in on.values()}
    # experts were seeded as copies of the layer's dense MLP (DeepSpeed MoE(expert=mlp) deep copy)
    sd = model.get_base_model().state_dict()
    k = "model.layers.1.mlp.deepspeed_moe.experts.deepspeed_experts.{}.down_proj.weight"
    assert torch.equal(sd[k.format(0)], sd[k.format(1)])


def test_trainable_flag_on_a_frozen_tensor_is_an_error():
    from model.LISA import LISAForCausalLM
    model = LISAForCausalLM(MedPLIBConfig.tiny(moe_enable=False, sam_depth=2), device="cpu")
    for n, p in model.named_parameters():
        p.requires_grad = "q_proj" in n
    with pytest.raises(NotImplementedError, match="frozen"):
        model.resolve_training_plan(model.parameters())


def test_trained_gate_reaches_the_model_without_a_forward():
    """merge.py's directory mode is build -> load checkpoint params -> sync_model -> merge_and_unload -> export with no
    forward in between; a trained `wg` (stage IV --sft_modules wg) must be in the exported state dict."""
    from model.MedPLIB import MedPLIBForCausalLM
    cfg = MedPLIBConfig.tiny(moe_enable=True, sam_depth=2)
    m = MedPLIBForCausalLM(cfg, device="cpu")
    lo = m.enable_lora(8, 16, 0.0, ("gate_proj", "up_proj", "down_proj"), train_gate=True, sft_modules=("wg",))
    name = "model.layers.1.mlp.deepspeed_moe.gate.wg.weight"
    with torch.no_grad():
        lo.full_param(name).add_(0.25)
    want = lo.full_param(name).detach().clone()
    lo.sync_model(m.model.llm)
    m.merge_and_unload()
    assert torch.equal(m.state_dict()[name].float(), want)
