"""ORACLE — TEST INFRASTRUCTURE ONLY (see oracle/ops.py header).

CPU fp32 restatement of the LLM side of the path from an HF-layout state dict (SURVEY §8b key names):
CLIP ViT-L vision tower + mm_projector, the multimodal splice, the Llama decoder stack with dense or DeepSpeed-MoE
MLPs, fp32 logits + filtered CE, and the <SEG> mask.

Pinning: the splice / seg-mask / TokenCompressor / MaskTokenEncoder functions restate reference code that is present in
/root/reference and are checked against the EXECUTED reference by oracle/make_golden.py (tests/golden/glue_reference.npz:
`LlavaMetaForCausalLM.prepare_inputs_labels_for_multimodal`, `MedPLIBForCausalLM.build_seg_token_mask`, the two modules).  Llama/CLIP arithmetic is third-party
(transformers==4.31.0, requirements.txt:137) — cross-checked in tests against the installed transformers' modules,
"parity unpinned" w.r.t. 4.31.0 itself.  DeepSpeed-MoE (deepspeed==0.13.1, requirements.txt:22) is absent from the
container: restated from SURVEY Appendix A.3, **parity unpinned**, guarded by known-answer tests."""
import math

import torch
import torch.nn.functional as F

from . import ops
from .ops import IGNORE_INDEX, IMAGE_TOKEN_INDEX

REGION_TOKEN_INDEX = -300          # utils/utils.py:9


# ----------------------------------------------------------------------------------------------- CLIP + projector
def clip_features(images, W, cfg, prefix="model.vision_tower.vision_tower.vision_model."):
    """HF CLIPVisionModel hidden_states[select_layer][:, 1:] (clip_encoder.py:31-60)."""
    C, H, p = cfg.clip_hidden_size, cfg.clip_num_heads, cfg.clip_patch_size
    x = F.conv2d(images, W[prefix + "embeddings.patch_embedding.weight"], stride=p).flatten(2).transpose(1, 2)
    cls = W[prefix + "embeddings.class_embedding"].expand(x.shape[0], 1, -1)
    x = torch.cat([cls, x], 1) + W[prefix + "embeddings.position_embedding.weight"]
    x = F.layer_norm(x, (C,), W[prefix + "pre_layrnorm.weight"], W[prefix + "pre_layrnorm.bias"], cfg.clip_ln_eps)
    hidden = [x]
    for i in range(cfg.clip_num_layers):
        lp = f"{prefix}encoder.layers.{i}."
        h = F.layer_norm(x, (C,), W[lp + "layer_norm1.weight"], W[lp + "layer_norm1.bias"], cfg.clip_ln_eps)
        q = F.linear(h, W[lp + "self_attn.q_proj.weight"], W[lp + "self_attn.q_proj.bias"])
        k = F.linear(h, W[lp + "self_attn.k_proj.weight"], W[lp + "self_attn.k_proj.bias"])
        v = F.linear(h, W[lp + "self_attn.v_proj.weight"], W[lp + "self_attn.v_proj.bias"])
        B, S, _ = q.shape
        a = ops.attention(q.view(B, S, H, -1), k.view(B, S, H, -1), v.view(B, S, H, -1))
        x = x + F.linear(a, W[lp + "self_attn.out_proj.weight"], W[lp + "self_attn.out_proj.bias"])
        h = F.layer_norm(x, (C,), W[lp + "layer_norm2.weight"], W[lp + "layer_norm2.bias"], cfg.clip_ln_eps)
        h = ops.quick_gelu(F.linear(h, W[lp + "mlp.fc1.weight"], W[lp + "mlp.fc1.bias"]))
        x = x + F.linear(h, W[lp + "mlp.fc2.weight"], W[lp + "mlp.fc2.bias"])
        hidden.append(x)
    return hidden[cfg.mm_vision_select_layer][:, 1:]


def mm_projector(x, W, prefix="model.mm_projector."):
    """mlp2x_gelu (multimodal_projector/builder.py:39-46)."""
    return F.linear(F.gelu(F.linear(x, W[prefix + "0.weight"], W[prefix + "0.bias"])), W[prefix + "2.weight"], W[prefix + "2.bias"])


def token_compressor(x, W, num_tokens, prefix="model.mm_token_compressor."):
    """TokenCompressor.forward (medplib_arch.py:67-77): AdaptiveAvgPool1d over tokens -> LayerNorm(eps 1e-5) -> Linear.
    x [n, 576, d] -> [n, num_tokens, d]."""
    d = x.shape[-1]
    x = F.adaptive_avg_pool1d(x.transpose(1, 2), num_tokens).transpose(1, 2)
    x = F.layer_norm(x, (d,), W[prefix + "norm.weight"], W[prefix + "norm.bias"], 1e-5)
    return F.linear(x, W[prefix + "proj.weight"], W[prefix + "proj.bias"])


def mask_token_encoder(masks, W, num_tokens, prefix="model.mask_encoder."):
    """MaskTokenEncoder.forward (medplib_arch.py:80-108): 4 x [Conv2d k3 s2 p1 + GELU] (1->64->128->256->256),
    flatten(2) -> AdaptiveAvgPool1d(num_tokens) -> transpose -> Linear(256, hidden) -> LayerNorm(hidden).
    masks [n, 1, H, W] (or [n, H, W]) -> [n, num_tokens, hidden]."""
    if masks.dim() == 3:
        masks = masks.unsqueeze(1)
    if masks.shape[1] != 1:
        masks = masks[:, :1]
    x = masks.to(W[prefix + "proj.weight"].dtype)
    for i in (0, 2, 4, 6):
        x = F.gelu(F.conv2d(x, W[prefix + f"encoder.{i}.weight"], W[prefix + f"encoder.{i}.bias"], stride=2, padding=1))
    x = F.adaptive_avg_pool1d(x.flatten(2), num_tokens).transpose(1, 2)
    x = F.linear(x, W[prefix + "proj.weight"], W[prefix + "proj.bias"])
    return F.layer_norm(x, (x.shape[-1],), W[prefix + "norm.weight"], W[prefix + "norm.bias"], 1e-5)


def init_icl_weights(hidden, seed, scale=0.1):
    """Seeded TokenCompressor / MaskTokenEncoder weights in the checkpoint key layout (model.mm_token_compressor.*,
    model.mask_encoder.*; medplib_arch.py:67-108).  Used by make_golden (loaded into the reference modules) and by the tests."""
    g = torch.Generator().manual_seed(seed)

    def rn(*shape, s=scale):
        return torch.randn(*shape, generator=g) * s
    W = {"model.mm_token_compressor.norm.weight": 1 + rn(hidden), "model.mm_token_compressor.norm.bias": rn(hidden),
         "model.mm_token_compressor.proj.weight": rn(hidden, hidden, s=hidden ** -0.5), "model.mm_token_compressor.proj.bias": rn(hidden)}
    cin = 1
    for i, cout in zip((0, 2, 4, 6), (64, 128, 256, 256)):
        W[f"model.mask_encoder.encoder.{i}.weight"] = rn(cout, cin, 3, 3, s=(cin * 9) ** -0.5)
        W[f"model.mask_encoder.encoder.{i}.bias"] = rn(cout)
        cin = cout
    W["model.mask_encoder.proj.weight"] = rn(hidden, 256, s=256 ** -0.5)
    W["model.mask_encoder.proj.bias"] = rn(hidden)
    W["model.mask_encoder.norm.weight"] = 1 + rn(hidden)
    W["model.mask_encoder.norm.bias"] = rn(hidden)
    return W


def combine_icl_features(image_features, mask_features, image_token_types):
    """medplib_arch.py:246-267: one feature block per placeholder, drawn from the image or the mask list by its type."""
    out, ii, mi = [], 0, 0
    for sample_types in image_token_types:
        for t in sample_types:
            if t == "mask":
                out.append(mask_features[mi]); mi += 1
            else:
                out.append(image_features[ii]); ii += 1
    return out


def extract_region_feature(region_feature_map, region_masks, max_sample_point, return_dtype=torch.float32):
    """medplib_arch.py:580-613 restated literally (+ rand_sample :33-39, point_sample :41-65).  region_feature_map [n, h*w, C]
    (region_fea_adapter output of the samples that have region masks), region_masks: per sample a list of [H, W] masks.
    Per mask: normalised (y, x) of the non-zero pixels (a random subset via torch.randperm when there are more than
    max_sample_point), bilinear grid_sample (align_corners=True) of the [C, h, w] map at those points, mean over the points."""
    out = []
    assert len(region_feature_map) == len(region_masks)
    for fmap, masks in zip(region_feature_map, region_masks):
        if len(masks) == 0:
            out.append(None)
            continue
        wh = torch.tensor([masks[0].shape[0], masks[0].shape[1]])[None,]
        pos = []
        for m in masks:
            x = m.nonzero() / wh
            if x.shape[0] > max_sample_point:
                x = x[torch.randperm(x.shape[0])[:max_sample_point], :]
            pos.append(x)
        pos = torch.nn.utils.rnn.pad_sequence(pos, padding_value=-1, batch_first=True)
        valid = ~(pos.sum(dim=-1) < 0)
        h = w = int(math.sqrt(fmap.shape[0]))
        c = fmap.shape[-1]
        dup = fmap.reshape(h, w, c).permute(2, 0, 1).unsqueeze(0).repeat(pos.shape[0], 1, 1, 1)
        grid = (2.0 * pos.flip(dims=(2,)).unsqueeze(2) - 1.0).float()
        samp = F.grid_sample(dup.float(), grid, align_corners=True).to(return_dtype).squeeze(3)      # [n_mask, C, P]
        samp = samp.to(dup.dtype)
        out.append(torch.stack([x[m].mean(dim=0) for x, m in zip(samp.transpose(1, 2), valid)]).nan_to_num())
    return out


# ----------------------------------------------------------------------------------------------- splice + seg mask
def prepare_inputs_labels_for_multimodal(input_ids, attention_mask, labels, image_features, embed_tokens, per_token=False,
                                         region_features=None, valid_region_masks_bool=None):
    """medplib_arch.py:296-527 restated literally (mm_use_im_start_end branch).
    image_features: [n_img, n_feat, d] (4-D images layout: one per sample, consumed per sample) or a flat list with one
    entry per placeholder (multi-image layouts).  region_features (optional): extract_region_feature's list over the samples
    with valid_region_masks_bool (bool per sample); REGION_TOKEN_INDEX ids after the last image are replaced by them (:409-433).
    Returns (attention_mask, inputs_embeds, labels)."""
    new_embeds, new_labels = [], [] if labels is not None else None
    cur_image_idx = 0
    for b, cur_ids in enumerate(input_ids):
        if (cur_ids == IMAGE_TOKEN_INDEX).sum() == 0:
            new_embeds.append(embed_tokens[cur_ids])
            if labels is not None:
                new_labels.append(labels[b])
            if not per_token:
                cur_image_idx += 1
            continue
        idx = torch.where(cur_ids == IMAGE_TOKEN_INDEX)[0]
        cur_e, cur_l = [], []
        cur_labels = labels[b] if labels is not None else None
        while idx.numel() > 0:
            feats = image_features[cur_image_idx]
            s = idx[0]
            cur_e.append(embed_tokens[cur_ids[:s]])
            cur_e.append(feats)
            cur_e.append(embed_tokens[cur_ids[s + 1:s + 2]])
            if labels is not None:
                cur_l.append(cur_labels[:s])
                cur_l.append(torch.full((feats.shape[0],), IGNORE_INDEX, dtype=labels.dtype))
                cur_l.append(cur_labels[s + 1:s + 2])
                cur_labels = cur_labels[s + 2:]
            cur_image_idx += 1
            cur_ids = cur_ids[s + 2:]
            idx = torch.where(cur_ids == IMAGE_TOKEN_INDEX)[0]
        if cur_ids.numel() > 0:
            region_indices = (cur_ids == REGION_TOKEN_INDEX).nonzero(as_tuple=True)[0].tolist()
            cur_ids = cur_ids[cur_ids != REGION_TOKEN_INDEX]
            text = embed_tokens[cur_ids]
            if labels is not None:
                cur_l.append(cur_labels)                       # labels keep the region positions (:420-421)
            if region_features is not None and valid_region_masks_bool[b] is not None:
                for k, indice in enumerate(region_indices):
                    ridx = int(torch.as_tensor(valid_region_masks_bool[:b + 1]).sum().item()) - 1
                    text = torch.cat((text[:indice], region_features[ridx][k].unsqueeze(0), text[indice:]))
            cur_e.append(text)
        new_embeds.append(torch.cat(cur_e, 0))
        if labels is not None:
            new_labels.append(torch.cat(cur_l, 0))
    L = input_ids.shape[1]
    max_len = max(x.shape[0] for x in new_embeds)
    emb = torch.stack([torch.cat([x, torch.zeros(max_len - x.shape[0], x.shape[1], dtype=x.dtype)], 0) for x in new_embeds])
    lab = None
    if labels is not None:
        lab = torch.stack([torch.cat([x, torch.full((max_len - x.shape[0],), IGNORE_INDEX, dtype=x.dtype)]) for x in new_labels])
    att = None
    if attention_mask is not None:
        rows = []
        for b in range(len(new_embeds)):
            n = new_embeds[b].shape[0]
            rows.append(torch.cat([torch.ones(n - L, dtype=torch.bool), attention_mask[b].bool(),
                                   torch.zeros(max_len - n, dtype=torch.bool)]))
        att = torch.stack(rows)
    return att, emb, lab


def build_seg_token_mask(input_ids, seg_token_idx, image_token_len, image_token_lengths=None):
    """model/MedPLIB.py:310-355 restated literally."""
    shifted = torch.zeros_like(input_ids, dtype=torch.bool)
    shifted[:, :-1] = input_ids[:, 1:] == seg_token_idx
    out = []
    for b, (ids, segs) in enumerate(zip(input_ids, shifted)):
        cur, k = [], 0
        for tok, is_seg in zip(ids, segs):
            if tok == IMAGE_TOKEN_INDEX:
                n = image_token_len
                if image_token_lengths is not None and len(image_token_lengths) > b and len(image_token_lengths[b]) > k:
                    n = image_token_lengths[b][k]
                k += 1
                cur.append(torch.zeros(n, dtype=torch.bool))
            else:
                cur.append(is_seg.view(1))
        out.append(torch.cat(cur))
    m = max(x.shape[0] for x in out)
    return torch.stack([torch.cat([x, torch.zeros(m - x.shape[0], dtype=torch.bool)]) for x in out])


# ----------------------------------------------------------------------------------------------- DeepSpeed MoE (top-1)
def top1_capacity_selection(mask1, capacity, rts_uniform=None):
    """The token-selection half of DeepSpeed 0.13.1 `top1gating`, as a function of the first choices alone:
    mask1 [T,E] one-hot of the gate argmax; with Random Token Selection the `capacity` largest entries per expert column of
    `mask1 * uniform` stay, the rest are dropped; the slot of a kept token is its rank among its expert's kept tokens in token
    order.  -> (new_mask1 [T,E], slot [T] (0 where dropped: mask with `kept`), kept [T] bool).  Also used by oracle/parity.py to
    check a routing kernel's kept / dropped sets at full size from that kernel's OWN expert choices."""
    T = mask1.shape[0]
    cap = min(capacity, T)
    if rts_uniform is None:
        # no draws supplied: stable first-come selection (documented deviation point; DeepSpeed's own behaviour without
        # RTS relies on topk tie-breaking)
        keep = (torch.cumsum(mask1, 0) - 1 < capacity) & mask1.bool()
        new_mask1 = keep.long()
    else:
        top_idx = torch.topk(mask1 * rts_uniform, k=cap, dim=0)[1]
        new_mask1 = mask1 * torch.zeros_like(mask1).scatter_(0, top_idx, 1)
    loc = torch.cumsum(new_mask1, 0) - 1
    slot = torch.sum(loc * new_mask1, 1)
    kept = new_mask1.sum(1).bool()
    return new_mask1, slot, kept


def moe_top1(x, wg, experts, capacity, rts_uniform=None):
    """DeepSpeed 0.13.1 TopKGate(k=1) + MOELayer on one rank — parity unpinned (third-party).
    x [T,d] fp32; wg [E,d]; experts: list of callables; rts_uniform [T,E] (the `uniform(mask1.shape)` draw) or None.
    Returns (out [T,d], l_aux, exp_counts [E], expert idx [T], slot [T] (-1 = dropped))."""
    T, E = x.shape[0], wg.shape[0]
    logits = x.float() @ wg.float().t()
    gates = F.softmax(logits, dim=1)
    idx = torch.argmax(gates, dim=1)
    mask1 = F.one_hot(idx, num_classes=E)
    exp_counts = mask1.sum(0)
    me, ce = gates.mean(0), mask1.float().mean(0)
    l_aux = torch.sum(me * ce) * E
    new_mask1, slot, kept = top1_capacity_selection(mask1, capacity, rts_uniform)
    gate_w = (gates * new_mask1.float()).sum(1)
    out = torch.zeros_like(x, dtype=torch.float32)
    for e in range(E):
        sel = kept & (idx == e)
        if sel.any():
            out[sel] = gate_w[sel, None] * experts[e](x[sel].float())
    slot = torch.where(kept, slot, torch.full_like(slot, -1))
    return out, l_aux, exp_counts, idx, slot


def moe_top2(x, wg, experts, capacity, noise=None):
    """DeepSpeed 0.13.1 top2gating + MOELayer on one rank (deepspeed/moe/sharded_moe.py — third-party,
    absent from the container: **parity unpinned**, restated literally from the published algorithm).
    noise [T,E]: the Gumbel draws of `top2_2nd_expert_sampling` (None = no sampling noise).
    Returns (out [T,d], l_aux, exp_counts [E], (idx1, idx2) [T], (slot1, slot2) [T] (-1 = dropped), (w1, w2))."""
    T, E = x.shape[0], wg.shape[0]
    logits = x.float() @ wg.float().t()
    gates = F.softmax(logits, dim=1)
    idx1 = torch.argmax(gates, dim=1)
    mask1 = F.one_hot(idx1, num_classes=E)
    lw = logits + noise if noise is not None else logits
    idx2 = torch.argmax(lw.masked_fill(mask1.bool(), float("-inf")), dim=1)
    mask2 = F.one_hot(idx2, num_classes=E)
    loc1 = torch.cumsum(mask1, 0) - 1
    loc2 = torch.cumsum(mask2, 0) - 1
    loc2 = loc2 + mask1.sum(0, keepdim=True)                 # second choices queue behind all first choices
    me, ce = gates.mean(0), mask1.float().mean(0)
    l_aux = torch.mean(me * ce) * E * E
    exp_counts = (mask1 + mask2).sum(0)
    mask1 = mask1 * (loc1 < capacity)
    mask2 = mask2 * (loc2 < capacity)
    slot1, slot2 = (loc1 * mask1).sum(1), (loc2 * mask2).sum(1)
    g1, g2 = (gates * mask1.float()).sum(1), (gates * mask2.float()).sum(1)
    den = torch.clamp(g1 + g2, min=torch.finfo(torch.float32).eps)
    g1, g2 = g1 / den, g2 / den
    k1, k2 = mask1.sum(1).bool(), mask2.sum(1).bool()
    out = torch.zeros_like(x, dtype=torch.float32)
    for e in range(E):
        for sel, g in ((k1 & (idx1 == e), g1), (k2 & (idx2 == e), g2)):
            if sel.any():
                out[sel] += g[sel, None] * experts[e](x[sel].float())
    slot1 = torch.where(k1, slot1, torch.full_like(slot1, -1)); slot2 = torch.where(k2, slot2, torch.full_like(slot2, -1))
    return out, l_aux, exp_counts, (idx1, idx2), (slot1, slot2), (g1, g2)


# ----------------------------------------------------------------------------------------------- Llama stack
def llama_forward(embeds, key_valid, W, cfg, training=True, rts=None, prefix="", collect=None):
    """MoELlamaModel_forward + MoELlamaDecoderLayer_forward (medplib_moe_llama.py:110-305) over HF-4.31 Llama modules
    (SURVEY A.1).  embeds [B,S,d]; key_valid bool [B,S] or None; rts: dict layer -> uniforms [T,E].
    Returns (final-normed hidden [B,S,d], [l_aux...])."""
    B, S, d = embeds.shape
    H, D, ff = cfg.num_attention_heads, cfg.head_dim, cfg.intermediate_size
    cos, sin = ops.rope_tables(S, D, cfg.rope_theta)
    moe_layers = cfg.moe_layer_set()
    x = embeds.float()
    aux = []
    for i in range(cfg.num_hidden_layers):
        p = f"{prefix}model.layers.{i}."
        h = ops.rmsnorm(x, W[p + "input_layernorm.weight"].float(), cfg.rms_norm_eps)
        def alin(t, name):
            # peft 0.10 LoRA Linear on an attention projection (parity unpinned): W x + (alpha / r) * B (A x)
            y = F.linear(t, W[p + f"self_attn.{name}.weight"])
            ka = p + f"self_attn.{name}.lora_A.default.weight"
            if ka in W:
                y = y + W["lora_scaling"] * F.linear(F.linear(t, W[ka]), W[p + f"self_attn.{name}.lora_B.default.weight"])
            return y
        q = alin(h, "q_proj").view(B, S, H, D)
        k = alin(h, "k_proj").view(B, S, H, D)
        v = alin(h, "v_proj").view(B, S, H, D)
        a = ops.attention(ops.rope(q, cos, sin), ops.rope(k, cos, sin), v, causal=True, key_valid=key_valid)
        x = x + alin(a, "o_proj")
        h = ops.rmsnorm(x, W[p + "post_attention_layernorm.weight"].float(), cfg.rms_norm_eps)
        if i in moe_layers:
            T = B * S
            cf = cfg.capacity_factor if training else cfg.eval_capacity_factor
            cap = max(int(math.ceil(T / cfg.num_experts * cf * cfg.top_k_experts)), cfg.min_capacity)   # top2gating: 2 * cf
            experts = []
            def elin(t, ep, name):
                # an expert's projection, with its peft LoRA adapter when the state dict carries one (stage IV; parity unpinned)
                y = F.linear(t, W[ep + name + ".weight"])
                ka = ep + name + ".lora_A.default.weight"
                if ka in W:
                    y = y + W["lora_scaling"] * F.linear(F.linear(t, W[ka]), W[ep + name + ".lora_B.default.weight"])
                return y
            for e in range(cfg.num_experts):
                ep = p + f"mlp.deepspeed_moe.experts.deepspeed_experts.{e}."
                experts.append(lambda t, ep=ep: elin(ops.swiglu(elin(t, ep, "gate_proj"), elin(t, ep, "up_proj")), ep, "down_proj"))
            if cfg.top_k_experts == 2:
                out, l_aux, counts, idx, slot, _ = moe_top2(h.reshape(T, d), W[p + "mlp.deepspeed_moe.gate.wg.weight"], experts, cap,
                                                            None if rts is None else rts.get(i))
            else:
                out, l_aux, counts, idx, slot = moe_top1(h.reshape(T, d), W[p + "mlp.deepspeed_moe.gate.wg.weight"], experts, cap,
                                                         None if rts is None else rts.get(i))
            aux.append(l_aux)
            if collect is not None:
                collect.append((idx, slot, counts))
            if getattr(cfg, "use_residual", False):
                # DeepSpeed MoE(use_residual=True).forward (deepspeed/moe/layer.py, 0.13.1 — third party, parity unpinned):
                # output_mlp = self.mlp(x); coef = softmax(self.coefficient(x), -1); out = out * coef[..., 0:1] + output_mlp * coef[..., 1:]
                hm = h.reshape(T, d)
                mlp = F.linear(ops.swiglu(F.linear(hm, W[p + "mlp.mlp.gate_proj.weight"]), F.linear(hm, W[p + "mlp.mlp.up_proj.weight"])),
                               W[p + "mlp.mlp.down_proj.weight"])
                coef = F.softmax(F.linear(hm, W[p + "mlp.coefficient.weight"], W[p + "mlp.coefficient.bias"]), dim=-1)
                out = out * coef[:, 0:1] + mlp * coef[:, 1:]
            x = x + out.view(B, S, d)
        else:
            def lin(t, name):
                # peft 0.10 LoRA Linear (parity unpinned: peft is not installed): W x + (alpha / r) * B (A x); dropout off here
                y = F.linear(t, W[p + f"mlp.{name}.weight"])
                ka = p + f"mlp.{name}.lora_A.default.weight"
                if ka in W:
                    y = y + W["lora_scaling"] * F.linear(F.linear(t, W[ka]), W[p + f"mlp.{name}.lora_B.default.weight"])
                return y
            m = lin(ops.swiglu(lin(h, "gate_proj"), lin(h, "up_proj")), "down_proj")
            x = x + m
    return ops.rmsnorm(x, W[prefix + "model.norm.weight"].float(), cfg.rms_norm_eps), aux


def causal_lm_loss(hidden, labels, W, cfg, aux, prefix=""):
    """medplib_moe_llama.py:388-421: fp32 logits, filtered shifted CE, + router_aux_loss_coef * sum(l_aux)."""
    logits = F.linear(hidden, W[prefix + "lm_head.weight"]).float()
    loss = ops.cross_entropy_filtered(logits, labels)
    if aux:
        loss = loss + cfg.router_aux_loss_coef * sum(aux)
    return loss, logits
