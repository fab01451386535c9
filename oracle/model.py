"""ORACLE — TEST INFRASTRUCTURE ONLY (see oracle/ops.py header).

End-to-end CPU fp32 restatement of `MedPLIBForCausalLM.model_forward` / `LISAForCausalLM.model_forward`
(model/MedPLIB.py:364-572, model/LISA.py:260-471) from an HF-layout state dict, plus seeded weight / batch generators
shared by the tests, smoke() and bench.py's cpu_baseline leg."""
import numpy as np
import contextlib

import torch
import torch.nn.functional as F

from . import llm, ops, sam


class UpcastDict(dict):
    """A weight dict whose bf16 entries read as fp32 (a fresh copy per access): DISTINCT decoder weights for all 32 layers at 7B dims are
    21.6 GB stored as the bf16 values they are (every such entry is bf16-representable by construction) instead of 43 GB of fp32; the oracle
    upcasts one matrix at a time."""

    def __getitem__(self, k):
        v = dict.__getitem__(self, k)
        return v.float() if torch.is_tensor(v) and v.dtype == torch.bfloat16 else v

    def get(self, k, default=None):
        return self[k] if k in self else default

    def items(self):
        return ((k, self[k]) for k in self.keys())


def init_hf_weights(cfg, seed=0, sam_seed=1234, store_bf16=False):
    """Seeded random weights in the HF checkpoint key layout, values rounded to bf16 where the product
    stores bf16 so both sides see identical numbers.  SAM-Med2D part: sam.init_weights under `model.visual_model.`.
    store_bf16: keep the decoder layers' matrices AS bf16 tensors in an UpcastDict (same values; see there)."""
    g = torch.Generator().manual_seed(seed)
    d, ff, V, C, I = cfg.hidden_size, cfg.intermediate_size, cfg.vocab_size, cfg.clip_hidden_size, cfg.clip_intermediate_size

    def rn(*shape, s=0.02, bf=True):
        t = torch.randn(*shape, generator=g) * s
        if bf and store_bf16 and len(shape) == 2 and shape[0] * shape[1] >= (1 << 20):
            return t.to(torch.bfloat16)
        return t.to(torch.bfloat16).float() if bf else t
    W = {"model.embed_tokens.weight": rn(V, d, s=0.5), "lm_head.weight": rn(V, d, s=0.05), "model.norm.weight": 1 + rn(d, s=0.1)}
    moe = cfg.moe_layer_set()
    for i in range(cfg.num_hidden_layers):
        p = f"model.layers.{i}."
        for n in ("q", "k", "v", "o"):
            W[p + f"self_attn.{n}_proj.weight"] = rn(d, d, s=1.0 / d ** 0.5)
        W[p + "input_layernorm.weight"] = 1 + rn(d, s=0.1)
        W[p + "post_attention_layernorm.weight"] = 1 + rn(d, s=0.1)
        if i in moe:
            W[p + "mlp.deepspeed_moe.gate.wg.weight"] = rn(cfg.num_experts, d, s=0.05, bf=False)
            for e in range(cfg.num_experts):
                ep = p + f"mlp.deepspeed_moe.experts.deepspeed_experts.{e}."
                W[ep + "gate_proj.weight"] = rn(ff, d, s=1.0 / d ** 0.5)
                W[ep + "up_proj.weight"] = rn(ff, d, s=1.0 / d ** 0.5)
                W[ep + "down_proj.weight"] = rn(d, ff, s=1.0 / ff ** 0.5)
            if getattr(cfg, "use_residual", False):
                W[p + "mlp.mlp.gate_proj.weight"] = rn(ff, d, s=1.0 / d ** 0.5)
                W[p + "mlp.mlp.up_proj.weight"] = rn(ff, d, s=1.0 / d ** 0.5)
                W[p + "mlp.mlp.down_proj.weight"] = rn(d, ff, s=1.0 / ff ** 0.5)
                W[p + "mlp.coefficient.weight"] = rn(2, d, s=0.3)
                W[p + "mlp.coefficient.bias"] = rn(2, s=0.3)
        else:
            W[p + "mlp.gate_proj.weight"] = rn(ff, d, s=1.0 / d ** 0.5)
            W[p + "mlp.up_proj.weight"] = rn(ff, d, s=1.0 / d ** 0.5)
            W[p + "mlp.down_proj.weight"] = rn(d, ff, s=1.0 / ff ** 0.5)
    tp = "model.vision_tower.vision_tower.vision_model."
    ps = cfg.clip_patch_size
    W[tp + "embeddings.patch_embedding.weight"] = rn(C, 3, ps, ps, s=0.03)
    W[tp + "embeddings.class_embedding"] = rn(C, s=0.5)
    W[tp + "embeddings.position_embedding.weight"] = rn(cfg.clip_num_patches + 1, C, s=0.1)
    W[tp + "pre_layrnorm.weight"] = 1 + rn(C, s=0.1); W[tp + "pre_layrnorm.bias"] = rn(C, s=0.1)
    for i in range(cfg.clip_num_layers):
        lp = f"{tp}encoder.layers.{i}."
        for n in ("layer_norm1", "layer_norm2"):
            W[lp + n + ".weight"] = 1 + rn(C, s=0.1); W[lp + n + ".bias"] = rn(C, s=0.1)
        for n in ("q_proj", "k_proj", "v_proj", "out_proj"):
            W[lp + f"self_attn.{n}.weight"] = rn(C, C, s=1.0 / C ** 0.5); W[lp + f"self_attn.{n}.bias"] = rn(C, s=0.05)
        W[lp + "mlp.fc1.weight"] = rn(I, C, s=1.0 / C ** 0.5); W[lp + "mlp.fc1.bias"] = rn(I, s=0.05)
        W[lp + "mlp.fc2.weight"] = rn(C, I, s=1.0 / I ** 0.5); W[lp + "mlp.fc2.bias"] = rn(C, s=0.05)
    W["model.mm_projector.0.weight"] = rn(d, C, s=1.0 / C ** 0.5); W["model.mm_projector.0.bias"] = rn(d, s=0.05)
    W["model.mm_projector.2.weight"] = rn(d, d, s=1.0 / d ** 0.5); W["model.mm_projector.2.bias"] = rn(d, s=0.05)
    W["model.region_fea_adapter.weight"] = rn(d, C, s=1.0 / C ** 0.5); W["model.region_fea_adapter.bias"] = rn(d, s=0.05)
    W["model.text_hidden_fcs.0.0.weight"] = rn(d, d, s=1.0 / d ** 0.5, bf=False); W["model.text_hidden_fcs.0.0.bias"] = rn(d, s=0.05, bf=False)
    W["model.text_hidden_fcs.0.2.weight"] = rn(cfg.out_dim, d, s=1.0 / d ** 0.5, bf=False)
    W["model.text_hidden_fcs.0.2.bias"] = rn(cfg.out_dim, s=0.05, bf=False)
    if getattr(cfg, "mm_token_compress", False) or getattr(cfg, "icl_mask_encoder", False):
        for k, v in llm.init_icl_weights(d, seed + 7).items():
            if ("mm_token_compressor" in k and cfg.mm_token_compress) or ("mask_encoder" in k and cfg.icl_mask_encoder):
                W[k] = v.to(torch.bfloat16).float()
    S = sam.init_weights(seed=sam_seed, encoder_depth=cfg.sam_depth)
    for k, v in S.items():
        if k.startswith("image_encoder.") and "rel_pos" not in k and "norm" not in k and ".bias" not in k and "channel" not in k \
                and "neck.1" not in k and "neck.3" not in k:
            v = v.to(torch.bfloat16).float()          # the product stores these GEMM operands in bf16
        W["model.visual_model." + k] = v
    return UpcastDict(W) if store_bf16 else W


def init_decoder_layer_weights(cfg, seed=3):
    """Seeded weights of a ONE-layer decoder at cfg's dims in the HF key layout (embed_tokens, layer 0 dense or MoE, final norm,
    lm_head), bf16-representable.  Shared by the true-dims layer tests and oracle/make_golden.py: golden_llama_layer."""
    g = torch.Generator().manual_seed(seed)
    d, ff, E = cfg.hidden_size, cfg.intermediate_size, cfg.num_experts

    def rn(*shape, s):
        return (torch.randn(*shape, generator=g) * s).to(torch.bfloat16).float()
    W = {"model.embed_tokens.weight": rn(cfg.vocab_size, d, s=0.5), "lm_head.weight": rn(cfg.vocab_size, d, s=0.05),
         "model.norm.weight": 1 + rn(d, s=0.1)}
    p = "model.layers.0."
    for n in ("q", "k", "v", "o"):
        W[p + f"self_attn.{n}_proj.weight"] = rn(d, d, s=d ** -0.5)
    W[p + "input_layernorm.weight"] = 1 + rn(d, s=0.1); W[p + "post_attention_layernorm.weight"] = 1 + rn(d, s=0.1)
    if cfg.moe_enable:
        W[p + "mlp.deepspeed_moe.gate.wg.weight"] = torch.randn(E, d, generator=g) * 0.05
        for e in range(E):
            ep = p + f"mlp.deepspeed_moe.experts.deepspeed_experts.{e}."
            W[ep + "gate_proj.weight"] = rn(ff, d, s=d ** -0.5); W[ep + "up_proj.weight"] = rn(ff, d, s=d ** -0.5)
            W[ep + "down_proj.weight"] = rn(d, ff, s=ff ** -0.5)
    else:
        W[p + "mlp.gate_proj.weight"] = rn(ff, d, s=d ** -0.5); W[p + "mlp.up_proj.weight"] = rn(ff, d, s=d ** -0.5)
        W[p + "mlp.down_proj.weight"] = rn(d, ff, s=ff ** -0.5)
    return W, g


def decoder_layer_inputs(cfg, g, B=2, S=639):
    """Inputs of the true-dims layer check, drawn from the generator init_decoder_layer_weights returns (same stream)."""
    emb = (torch.randn(B, S, cfg.hidden_size, generator=g) * 0.5).to(torch.bfloat16)
    kv = torch.ones(B, S, dtype=torch.bool); kv[1, 600:] = False
    return emb, kv


def init_hf_weights_aliased(cfg, seed=0):
    """True-dims weights with bounded host memory: ONE decoder layer's seeded weights referenced by all `cfg.num_hidden_layers`
    layers (the same tensor objects: 1.6 GB of fp32 instead of 52 GB for 32 layers; arithmetic and memory traffic per layer are what
    distinct weights would cost).  Used by bench.py's cpu_baseline leg and the full-depth parity check (oracle/parity.py)."""
    import copy
    moe_set = cfg.moe_layer_set()
    mixed = 0 < len(moe_set) < cfg.num_hidden_layers          # dense AND MoE layers (moe_mode second_half / sparse: the reference's default)
    cfg1 = copy.deepcopy(cfg)
    cfg1.num_hidden_layers = 2 if mixed else 1
    if mixed:
        cfg1.moe_layers_idx = [1]                               # prototype layer 0 = dense, 1 = MoE
    elif cfg1.moe_layers_idx is not None:
        cfg1.moe_layers_idx = [0] if 0 in cfg.moe_layers_idx else []
    W1 = init_hf_weights(cfg1, seed=seed)
    W = {k: v for k, v in W1.items() if not k.startswith("model.layers.")}
    for i in range(cfg.num_hidden_layers):
        src = f"model.layers.{1 if (mixed and i in moe_set) else 0}."
        for k in [k for k in W1 if k.startswith(src)]:
            W[f"model.layers.{i}." + k[len(src):]] = W1[k]
    return W


def make_batch(cfg, B, L=64, H=96, Wd=80, seed=0, ragged=False, sam_size=256):
    """Synthetic batch in the collator's contract (datasets/DataCollatorForSupervisedDataset.py:11-138):
    one IMAGE placeholder bracketed by im_start/im_end, <SEG> near the end, labels supervised on the tail."""
    g = torch.Generator().manual_seed(seed)
    V = cfg.vocab_size
    ids = torch.randint(3, min(V, cfg.seg_token_idx) - 1, (B, L), generator=g)
    img_pos = min(35, L // 2)
    labels = torch.full((B, L), ops.IGNORE_INDEX, dtype=torch.int64)
    att = torch.ones(B, L, dtype=torch.bool)
    for b in range(B):
        npad = ((b * 3) % 5) if ragged else 0
        n = L - npad                                  # real tokens; right padding like the collator (pad id 0, label -100)
        ids[b, 0] = 1
        ids[b, img_pos - 1], ids[b, img_pos], ids[b, img_pos + 1] = V - 2, ops.IMAGE_TOKEN_INDEX, V - 1   # <im_start> <image> <im_end>
        ids[b, n - 3] = cfg.seg_token_idx
        ids[b, n - 1] = 2
        ids[b, n:] = 0
        labels[b, n - 8:n] = ids[b, n - 8:n]
        att[b, n:] = False
    images = torch.randn(B, 3, sam_size, sam_size, generator=g)
    images_clip = torch.randn(B, 3, cfg.clip_image_size, cfg.clip_image_size, generator=g)
    masks = []
    yy, xx = torch.meshgrid(torch.arange(H), torch.arange(Wd), indexing="ij")
    for b in range(B):
        cy, cx = torch.rand(2, generator=g) * torch.tensor([H, Wd])
        r = 8 + torch.rand(1, generator=g) * min(H, Wd) / 3
        masks.append((((yy - cy) ** 2 + (xx - cx) ** 2) < r ** 2).float())
    return {"images": images, "images_clip": images_clip, "input_ids": ids, "labels": labels, "attention_mask": att,
            "masks_list": masks, "label_list": [torch.full((H, Wd), 255.0) for _ in range(B)],
            "resize_list": [(sam_size, sam_size)] * B, "valid_mask_bool": [[True]] * B, "offset": None, "region_masks": [],
            "inference": False, "seg_flag": True}


def make_batch_icl(cfg, B, n_ctx=2, H=96, Wd=80, seed=0, sam_size=256, mask_size=64):
    """ICL separate-mode batch (BASELINE config 5 shape, datasets/ICLLazySupervisedDataset.py + collator :105-108): per sample
    n_ctx in-context (image, mask) pairs + the query image = 2*n_ctx+1 placeholders, `image_token_types` [image, mask]*n_ctx +
    [image], `image_token_lengths` per placeholder, one <SEG> per sample."""
    g = torch.Generator().manual_seed(seed)
    V = cfg.vocab_size
    n_ph = 2 * n_ctx + 1
    L = 8 + 4 * n_ph + 12
    ids = torch.randint(3, min(V, cfg.seg_token_idx) - 1, (B, L), generator=g)
    labels = torch.full((B, L), ops.IGNORE_INDEX, dtype=torch.int64)
    att = torch.ones(B, L, dtype=torch.bool)
    for b in range(B):
        ids[b, 0] = 1
        for k in range(n_ph):
            p = 6 + 4 * k
            ids[b, p - 1], ids[b, p], ids[b, p + 1] = V - 2, ops.IMAGE_TOKEN_INDEX, V - 1
        ids[b, L - 3] = cfg.seg_token_idx
        ids[b, L - 1] = 2
        labels[b, L - 8:] = ids[b, L - 8:]
    tok = cfg.mm_compressed_token_count if cfg.mm_token_compress else cfg.clip_num_patches
    types = [["image", "mask"] * n_ctx + ["image"] for _ in range(B)]
    lengths = [[tok, cfg.mask_encoder_token_count] * n_ctx + [tok] for _ in range(B)]
    images_clip = [torch.randn(n_ctx + 1, 3, cfg.clip_image_size, cfg.clip_image_size, generator=g) for _ in range(B)]
    mask_images = [(torch.rand(n_ctx, 1, mask_size, mask_size, generator=g) > 0.5).float() for _ in range(B)]
    base = make_batch(cfg, B, L=64, H=H, Wd=Wd, seed=seed + 1, sam_size=sam_size)
    base.update(input_ids=ids, labels=labels, attention_mask=att, images_clip=images_clip, mask_images=mask_images,
                image_token_types=types, image_token_lengths=lengths, icl_image_counts=[n_ctx + 1] * B)
    return base


def expand_embedding(image_embeddings, valid_mask_bool):
    """`expand_embedding` (model/MedPLIB.py:292-308, model/LISA.py:228-239): image embedding i repeated once per GT mask of sample
    i, samples without a mask DROPPED; unchanged when `valid_mask_bool` is empty."""
    if valid_mask_bool is None or len(valid_mask_bool) == 0:
        return image_embeddings
    parts = [image_embeddings[i:i + 1].expand(len(m), -1, -1, -1) for i, m in enumerate(valid_mask_bool) if m]
    return torch.cat(parts, 0)


def make_batch_multimask(cfg, L=64, seed=0, sam_size=256, sizes=((96, 80), (64, 72), (96, 80))):
    """Three samples with valid_mask_bool = [[True], [True, True], []] (the collator's flat per-mask lists,
    datasets/DataCollatorForSupervisedDataset.py:31-50): sample 0 one <SEG> + one mask, sample 1 two <SEG> + two masks (of different
    sizes), sample 2 a VQA sample without <SEG> / masks whose image embedding expand_embedding drops."""
    b = make_batch(cfg, 3, L=L, seed=seed, ragged=True, sam_size=sam_size)
    ids, labels = b["input_ids"], b["labels"]
    n1 = int(b["attention_mask"][1].sum())
    ids[1, n1 - 6] = cfg.seg_token_idx                     # second <SEG> of sample 1 (the first sits at n1 - 3)
    labels[1, n1 - 8:n1] = ids[1, n1 - 8:n1]
    n2 = int(b["attention_mask"][2].sum())
    g = torch.Generator().manual_seed(seed + 991)
    ids[2, n2 - 3] = int(torch.randint(3, cfg.seg_token_idx - 1, (1,), generator=g))     # sample 2: no <SEG>
    labels[2, n2 - 8:n2] = ids[2, n2 - 8:n2]
    masks = []
    for (H, Wd) in sizes:
        yy, xx = torch.meshgrid(torch.arange(H), torch.arange(Wd), indexing="ij")
        cy, cx = torch.rand(2, generator=g) * torch.tensor([H, Wd])
        r = 8 + torch.rand(1, generator=g) * min(H, Wd) / 3
        masks.append((((yy - cy) ** 2 + (xx - cx) ** 2) < r ** 2).float())
    b.update(masks_list=masks, label_list=[torch.full(s_, 255.0) for s_ in sizes], resize_list=[(sam_size, sam_size)] * 3,
             valid_mask_bool=[[True], [True, True], []])
    return b


def model_forward(batch, W, cfg, training=True, rts=None, return_intermediates=False, override=None, llm_grad=False, collect=None):
    """model/MedPLIB.py:364-572 end to end on the CPU in fp32.  `override` (tests only) may inject `hidden` [B,S,d],
    `image_emb` [B,256,16,16] and `ce` so the trainable tail can be checked on exactly the trunk outputs another
    implementation produced."""
    ids, labels, att = batch["input_ids"], batch["labels"], batch["attention_mask"]
    B = ids.shape[0]
    # llm_grad (training tests): the front end stays on the autograd tape too (trainable projector / compressor / embed_tokens)
    with (torch.enable_grad() if llm_grad else torch.no_grad()):
        image_emb = sam.image_encoder(batch["images"], {k[len("model.visual_model."):]: v for k, v in W.items()
                                                        if k.startswith("model.visual_model.")}, depth=cfg.sam_depth)
        clip_in = batch["images_clip"]
        multi = isinstance(clip_in, (list, tuple)) or clip_in.dim() == 5
        raw_feats = llm.clip_features(torch.cat(list(clip_in), 0) if multi else clip_in, W, cfg)
        with (torch.enable_grad() if llm_grad else contextlib.nullcontext()):      # mm_projector may be trainable (--sft_modules)
            feats = llm.mm_projector(raw_feats, W)
        tok = cfg.clip_num_patches
        if getattr(cfg, "mm_token_compress", False):                    # encode_images, medplib_arch.py:198-202
            tok = cfg.mm_compressed_token_count
            with (torch.enable_grad() if llm_grad else contextlib.nullcontext()):  # mm_token_compressor may be trainable
                feats = llm.token_compressor(feats, W, tok)
        feat_list, per_token = feats, False
        if batch.get("image_token_types") is not None and batch.get("mask_images") is not None and len(batch["mask_images"]) > 0:
            mf = llm.mask_token_encoder(torch.cat(list(batch["mask_images"]), 0), W, cfg.mask_encoder_token_count)
            feat_list, per_token = llm.combine_icl_features(list(feats), list(mf), batch["image_token_types"]), True
        elif multi:
            feat_list, per_token = list(feats), True
        region_features = valid = None
        if batch.get("region_masks") is not None and len(batch["region_masks"]) > 0:        # medplib_arch.py:221-227, 283-295
            valid = torch.tensor([any(v) for v in batch["valid_region_masks_bool"]])
            rmap = F.linear(raw_feats, W["model.region_fea_adapter.weight"], W["model.region_fea_adapter.bias"])[valid]
            region_features = llm.extract_region_feature(rmap, batch["region_masks"], cfg.max_sample_point)
        with (torch.enable_grad() if llm_grad else contextlib.nullcontext()):      # embed_tokens may be trainable (--sft_modules)
            att2, embeds, lab2 = llm.prepare_inputs_labels_for_multimodal(ids, att, labels, feat_list, W["model.embed_tokens.weight"], per_token,
                                                                          region_features=region_features, valid_region_masks_bool=valid)
        kv = None if att2.all() else att2
    # llm_grad (LoRA training): the decoder and the CE stay on the autograd tape so `loss.backward()` reaches the adapters in W
    with (contextlib.nullcontext() if llm_grad else torch.no_grad()):
        hidden, aux = llm.llama_forward(embeds, kv, W, cfg, training=training, rts=rts, collect=collect)
        ce, logits = llm.causal_lm_loss(hidden, lab2, W, cfg, aux)
        if override:
            hidden = override.get("hidden", hidden); image_emb = override.get("image_emb", image_emb); ce = override.get("ce", ce)
    seg_mask = llm.build_seg_token_mask(ids, cfg.seg_token_idx, tok, batch.get("image_token_lengths"))
    SW = {k[len("model.visual_model."):]: v for k, v in W.items() if k.startswith("model.visual_model.")}
    hid = hidden if llm_grad else hidden.detach()
    fc = lambda x: F.linear(F.relu(F.linear(x, W["model.text_hidden_fcs.0.0.weight"], W["model.text_hidden_fcs.0.0.bias"])),
                            W["model.text_hidden_fcs.0.2.weight"], W["model.text_hidden_fcs.0.2.bias"])
    last = fc(hid)                                            # applied to every row like the reference (MedPLIB.py:456)
    pred_emb = last[seg_mask]
    if batch.get("icl_image_counts") is not None and len(batch["masks_list"]) > 0:
        pred_emb = pred_emb[-len(batch["masks_list"]):]          # MedPLIB.py:462-463
    pe = sam.dense_pe(SW)
    image_emb = expand_embedding(image_emb, batch.get("valid_mask_bool"))
    pred_masks, pred_ious, low = [], [], []
    for i in range(len(pred_emb)):                 # pred_emb[i] is paired with image_emb[i] BY POSITION (MedPLIB.py:473-487)
        sp, de = sam.prompt_encoder_text(pred_emb[i].view(1, 1, -1), SW)
        lm, io = sam.mask_decoder(image_emb[i:i + 1], pe, sp, de, SW)
        low.append(lm)
        pm = ops.postprocess_masks(lm, batch["resize_list"][i], tuple(batch["label_list"][i].shape))
        pred_masks.append(pm[:, 0]); pred_ious.append(io[:, 0])
    weights = dict(ce=cfg.ce_loss_weight, bce=cfg.bce_loss_weight, dice=cfg.dice_loss_weight, iou=cfg.iou_loss_weight,
                   focal=cfg.focal_loss_weight)
    out = ops.combine_mask_losses(pred_masks, batch["masks_list"], pred_ious, ce, weights)
    if return_intermediates:
        return out, dict(hidden=hidden, ce=ce, seg_mask=seg_mask, pred_emb=pred_emb, low_res=torch.cat(low), pred_masks=pred_masks,
                         pred_ious=torch.cat(pred_ious), image_emb=image_emb, feats=feats, embeds=embeds, labels=lab2, att=att2)
    return out


def evaluate(batch, W, cfg, max_new_tokens=8, eos_token_id=2, return_debug=False):
    """`MedPLIBForCausalLM.evaluate` (model/MedPLIB.py:574-680) on the CPU in fp32: greedy decoding (HF generate, do_sample=False)
    restated without a KV cache (each step re-runs the causal stack on the grown sequence — identical values), the
    concatenated last-layer hidden states, <SEG> pick rules and one mask.  B = 1."""
    ids = batch["input_ids"]
    assert ids.shape[0] == 1
    with torch.no_grad():
        feats = llm.mm_projector(llm.clip_features(batch["images_clip"], W, cfg), W)
        _, embeds, _ = llm.prepare_inputs_labels_for_multimodal(ids, None, None, feats, W["model.embed_tokens.weight"])
        seq = embeds
        generated, gaps = [], []
        for _ in range(max_new_tokens):
            hidden, _ = llm.llama_forward(seq, None, W, cfg, training=False)
            logits = F.linear(hidden[:, -1], W["lm_head.weight"]).float()
            top2 = torch.topk(logits[0], 2).values
            gaps.append(float(top2[0] - top2[1]))
            tok = int(torch.argmax(logits, -1))
            generated.append(tok)
            if tok == eos_token_id or len(generated) == max_new_tokens:
                break
            seq = torch.cat([seq, W["model.embed_tokens.weight"][tok].view(1, 1, -1)], 1)
        output_ids = torch.cat([ids, torch.tensor([generated])], 1)
        n_hidden = hidden.shape[1]
        seg_mask = llm.build_seg_token_mask(output_ids, cfg.seg_token_idx, cfg.clip_num_patches)[:, :n_hidden]
        last = F.linear(F.relu(F.linear(hidden, W["model.text_hidden_fcs.0.0.weight"], W["model.text_hidden_fcs.0.0.bias"])),
                        W["model.text_hidden_fcs.0.2.weight"], W["model.text_hidden_fcs.0.2.bias"])
        pred = last[seg_mask]
        if pred.shape[0] > 1:
            pred = pred[:1]
        elif pred.shape[0] == 0:
            pred = last[:1, -2:-1, :].squeeze(1)
        SW = {k[len("model.visual_model."):]: v for k, v in W.items() if k.startswith("model.visual_model.")}
        image_emb = sam.image_encoder(batch["images"], SW, depth=cfg.sam_depth)
        sp, de = sam.prompt_encoder_text(pred[0].view(1, 1, -1), SW)
        lm, _ = sam.mask_decoder(image_emb[:1], sam.dense_pe(SW), sp, de, SW)
        pm = ops.postprocess_masks(lm, batch["resize_list"][0], tuple(batch["label_list"][0].shape))
    if return_debug:
        return output_ids, [pm[:, 0]], dict(gaps=gaps, hidden=hidden)
    return output_ids, [pm[:, 0]]
