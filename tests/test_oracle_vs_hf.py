"""CPU: the oracle's Llama / CLIP arithmetic against the installed HuggingFace modules.

The reference pins transformers==4.31.0 (requirements.txt:137), which is not in the image; 5.15.0 is, and implements the same
math for these two models (RMSNorm, half-split RoPE, SwiGLU, causal softmax attention; CLIP ViT with quick_gelu, pre-LN,
hidden_states[-2]).  These tests load the oracle's seeded HF-layout weights into the HF modules and compare outputs — a cross-
check of the restatement, not a pin on 4.31.0 itself."""
import pytest
import torch

from medplib_amd.model.config import MedPLIBConfig
from oracle import llm as OL
from oracle import model as OM


def test_oracle_llama_matches_hf_llama():
    from transformers import LlamaConfig, LlamaModel
    cfg = MedPLIBConfig.tiny(moe_enable=False, num_hidden_layers=3)
    W = OM.init_hf_weights(cfg)
    hc = LlamaConfig(vocab_size=cfg.vocab_size, hidden_size=cfg.hidden_size, intermediate_size=cfg.intermediate_size,
                     num_hidden_layers=cfg.num_hidden_layers, num_attention_heads=cfg.num_attention_heads,
                     num_key_value_heads=cfg.num_attention_heads, rms_norm_eps=cfg.rms_norm_eps, rope_theta=cfg.rope_theta,
                     max_position_embeddings=cfg.max_position_embeddings, attention_bias=False, hidden_act="silu")
    hc._attn_implementation = "eager"
    hf = LlamaModel(hc).eval()
    sd = {k[len("model."):]: v for k, v in W.items() if k.startswith("model.layers.") or k in ("model.norm.weight", "model.embed_tokens.weight")}
    missing, unexpected = hf.load_state_dict(sd, strict=False)
    assert not unexpected and all("rotary" in m or "inv_freq" in m for m in missing), (missing, unexpected)
    g = torch.Generator().manual_seed(0)
    B, S = 2, 37
    emb = torch.randn(B, S, cfg.hidden_size, generator=g) * 0.5
    att = torch.ones(B, S, dtype=torch.long); att[1, 30:] = 0            # right padding: positions stay arange(S)
    pos = torch.arange(S)[None].expand(B, -1)
    with torch.no_grad():
        ref = hf(inputs_embeds=emb, attention_mask=att, position_ids=pos).last_hidden_state
        got, _ = OL.llama_forward(emb, att.bool(), W, cfg, training=False)
    valid = att.bool()
    assert (got[valid] - ref[valid]).abs().max().item() < 2e-4, (got[valid] - ref[valid]).abs().max()


def test_oracle_clip_matches_hf_clip():
    from transformers import CLIPVisionConfig, CLIPVisionModel
    cfg = MedPLIBConfig.tiny()
    W = OM.init_hf_weights(cfg)
    hc = CLIPVisionConfig(hidden_size=cfg.clip_hidden_size, intermediate_size=cfg.clip_intermediate_size,
                          num_hidden_layers=cfg.clip_num_layers, num_attention_heads=cfg.clip_num_heads, image_size=cfg.clip_image_size,
                          patch_size=cfg.clip_patch_size, hidden_act="quick_gelu", layer_norm_eps=cfg.clip_ln_eps)
    hc._attn_implementation = "eager"
    hf = CLIPVisionModel(hc).eval()
    tp = "model.vision_tower.vision_tower."
    sd = {k[len(tp):]: v for k, v in W.items() if k.startswith(tp)}
    if not any(k.startswith("vision_model.") for k in hf.state_dict()):      # 5.x flattened the wrapper: keys lost the prefix
        sd = {k[len("vision_model."):]: v for k, v in sd.items()}
    missing, unexpected = hf.load_state_dict(sd, strict=False)
    assert not unexpected and all(("post_layernorm" in m) or ("position_ids" in m) for m in missing), (missing, unexpected)
    g = torch.Generator().manual_seed(1)
    img = torch.randn(2, 3, cfg.clip_image_size, cfg.clip_image_size, generator=g)
    with torch.no_grad():
        hs = hf(pixel_values=img, output_hidden_states=True).hidden_states
        ref = hs[cfg.mm_vision_select_layer][:, 1:]                      # feature_select 'patch' (clip_encoder.py:31-39)
        got = OL.clip_features(img, W, cfg)
    assert (got - ref).abs().max().item() < 2e-4, (got - ref).abs().max()
