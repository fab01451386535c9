"""ORACLE — TEST INFRASTRUCTURE ONLY (see oracle/ops.py header).

Functional CPU fp32 restatement of SAM-Med2D as MedPLIB uses it (image 256 px, ViT-B + adapters, text-prompt-only
prompt encoder, mask decoder with multimask_output=False).  Weights come in a flat dict with the REFERENCE's
state-dict key names (`image_encoder.*`, `prompt_encoder.*`, `mask_decoder.*`).  Pinned against the reference modules
(imported from /root/reference in the dev container) by oracle/make_golden.py -> tests/golden/sam_*.npz."""
import math

import numpy as np
import torch
import torch.nn.functional as F

from . import ops

GLOBAL_ATTN = (2, 5, 8, 11)   # build_sam.py:56
WINDOW = 14                   # build_sam.py:99
HEADS = 12
LN_EPS = 1e-6                 # build_sam.py:91


# ----------------------------------------------------------------------------------------------- image encoder
def _ln(x, W, pre, eps=LN_EPS):
    return F.layer_norm(x, (x.shape[-1],), W[pre + ".weight"], W[pre + ".bias"], eps)


def _ln2d(x, W, pre, eps=1e-6):
    """modeling/common.py:31-45 on NCHW."""
    u = x.mean(1, keepdim=True)
    s = (x - u).pow(2).mean(1, keepdim=True)
    x = (x - u) / torch.sqrt(s + eps)
    return W[pre + ".weight"][:, None, None] * x + W[pre + ".bias"][:, None, None]


def window_partition(x, ws):
    """image_encoder.py:299-320."""
    B, H, Wd, C = x.shape
    ph, pw = (ws - H % ws) % ws, (ws - Wd % ws) % ws
    if ph or pw:
        x = F.pad(x, (0, 0, 0, pw, 0, ph))
    Hp, Wp = H + ph, Wd + pw
    x = x.view(B, Hp // ws, ws, Wp // ws, ws, C)
    return x.permute(0, 1, 3, 2, 4, 5).contiguous().view(-1, ws, ws, C), (Hp, Wp)


def window_unpartition(win, ws, pad_hw, hw):
    """image_encoder.py:323-345."""
    Hp, Wp = pad_hw
    H, Wd = hw
    B = win.shape[0] // (Hp * Wp // ws // ws)
    x = win.view(B, Hp // ws, Wp // ws, ws, ws, -1).permute(0, 1, 3, 2, 4, 5).contiguous().view(B, Hp, Wp, -1)
    return x[:, :H, :Wd, :].contiguous()


def encoder_attention(x, W, pre):
    """image_encoder.py:280-296 (Attention.forward with use_rel_pos)."""
    B, H, Wd, C = x.shape
    qkv = F.linear(x, W[pre + ".qkv.weight"], W[pre + ".qkv.bias"]).reshape(B, H * Wd, 3, HEADS, -1).permute(2, 0, 3, 1, 4)
    q, k, v = qkv.reshape(3, B * HEADS, H * Wd, -1).unbind(0)
    scale = q.shape[-1] ** -0.5
    attn = (q * scale) @ k.transpose(-2, -1)
    rel_h, rel_w = ops.decomposed_rel_pos(q, W[pre + ".rel_pos_h"], W[pre + ".rel_pos_w"], (H, Wd))
    attn = (attn.view(-1, H, Wd, H, Wd) + rel_h.view(-1, H, Wd, H)[..., None] + rel_w.view(-1, H, Wd, Wd)[:, :, :, None, :])
    attn = attn.view(-1, H * Wd, H * Wd).softmax(-1)
    x = (attn @ v).view(B, HEADS, H, Wd, -1).permute(0, 2, 3, 1, 4).reshape(B, H, Wd, -1)
    return F.linear(x, W[pre + ".proj.weight"], W[pre + ".proj.bias"])


def adapter(x, W, pre):
    """image_encoder.py:18-56 (Adapter_Layer): x is the norm2 output, NHWC."""
    x = x.permute(0, 3, 1, 2)
    B, C = x.shape[:2]
    pooled = x.mean((2, 3))
    ch = torch.sigmoid(F.linear(F.relu(F.linear(pooled, W[pre + ".channel.0.weight"])), W[pre + ".channel.2.weight"]))
    xc = ch.view(B, C, 1, 1) * x
    sp = F.relu(F.conv2d(xc, W[pre + ".spatial.0.weight"], stride=2, padding=1))
    sp = F.relu(F.conv_transpose2d(sp, W[pre + ".spatial.2.weight"], stride=2, padding=1))
    x = (x + sp).permute(0, 2, 3, 1)          # skip connects the UNscaled input (image_encoder.py:50-51)
    return _ln(x, W, pre + ".norm", eps=1e-5)   # Adapter_Layer builds a plain nn.LayerNorm (image_encoder.py:19-23)


def image_encoder(images, W, pre="image_encoder", depth=12, return_blocks=False):
    """image_encoder.py:151-162 + Block.forward :217-238.  images [B,3,256,256] -> [B,256,16,16]."""
    x = F.conv2d(images, W[pre + ".patch_embed.proj.weight"], W[pre + ".patch_embed.proj.bias"], stride=16).permute(0, 2, 3, 1)
    x = x + W[pre + ".pos_embed"]
    blocks = []
    for i in range(depth):
        bp = f"{pre}.blocks.{i}"
        shortcut = x
        h = _ln(x, W, bp + ".norm1")
        if i not in GLOBAL_ATTN:
            Hh, Ww = h.shape[1], h.shape[2]
            h, pad_hw = window_partition(h, WINDOW)
        h = encoder_attention(h, W, bp + ".attn")
        if i not in GLOBAL_ATTN:
            h = window_unpartition(h, WINDOW, pad_hw, (Hh, Ww))
        x = shortcut + h
        xn = _ln(x, W, bp + ".norm2")
        mlp = F.linear(F.gelu(F.linear(xn, W[bp + ".mlp.lin1.weight"], W[bp + ".mlp.lin1.bias"])),
                       W[bp + ".mlp.lin2.weight"], W[bp + ".mlp.lin2.bias"])
        x = x + mlp + adapter(xn, W, bp + ".Adapter")
        if return_blocks:
            blocks.append(x)
    y = x.permute(0, 3, 1, 2)
    y = F.conv2d(y, W[pre + ".neck.0.weight"])
    y = _ln2d(y, W, pre + ".neck.1")
    y = F.conv2d(y, W[pre + ".neck.2.weight"], padding=1)
    y = _ln2d(y, W, pre + ".neck.3")
    return (y, blocks) if return_blocks else y


# ----------------------------------------------------------------------------------------------- prompt encoder
def dense_pe(W, size=(16, 16), pre="prompt_encoder"):
    """prompt_encoder.py:62-71,204-226: cat(sin,cos)(2*pi*(2*xy-1) @ G), fp32, [1,256,h,w]."""
    h, w = size
    G = W[pre + ".pe_layer.positional_encoding_gaussian_matrix"].float()
    grid = torch.ones((h, w), dtype=torch.float32)
    y = (grid.cumsum(0) - 0.5) / h
    x = (grid.cumsum(1) - 0.5) / w
    c = 2 * torch.stack([x, y], -1) - 1
    c = 2 * np.pi * (c @ G)
    return torch.cat([torch.sin(c), torch.cos(c)], -1).permute(2, 0, 1).unsqueeze(0)


def prompt_encoder_text(text_embeds, W, size=(16, 16), pre="prompt_encoder"):
    """prompt_encoder.py:140-187 with points=boxes=masks=None: sparse = text_embeds [B,1,256];
    dense = no_mask_embed broadcast [B,256,h,w]."""
    B = text_embeds.shape[0]
    dense = W[pre + ".no_mask_embed.weight"].reshape(1, -1, 1, 1).expand(B, -1, size[0], size[1])
    return text_embeds, dense


# ----------------------------------------------------------------------------------------------- mask decoder
def _dec_attn(q, k, v, W, pre, heads=8):
    """transformer.py:218-244."""
    q = F.linear(q, W[pre + ".q_proj.weight"], W[pre + ".q_proj.bias"])
    k = F.linear(k, W[pre + ".k_proj.weight"], W[pre + ".k_proj.bias"])
    v = F.linear(v, W[pre + ".v_proj.weight"], W[pre + ".v_proj.bias"])

    def sep(x):
        b, n, c = x.shape
        return x.reshape(b, n, heads, c // heads).transpose(1, 2)
    q, k, v = sep(q), sep(k), sep(v)
    attn = torch.softmax(q @ k.permute(0, 1, 3, 2) / math.sqrt(q.shape[-1]), -1)
    out = (attn @ v).transpose(1, 2)
    out = out.reshape(out.shape[0], out.shape[1], -1)
    return F.linear(out, W[pre + ".out_proj.weight"], W[pre + ".out_proj.bias"])


def _dln(x, W, pre):
    return F.layer_norm(x, (x.shape[-1],), W[pre + ".weight"], W[pre + ".bias"], 1e-5)


def two_way_transformer(src, pos, tokens, W, pre="mask_decoder.transformer"):
    """transformer.py:62-106 and :151-182.  src/pos [B,C,h,w], tokens [B,N,C]."""
    keys = src.flatten(2).permute(0, 2, 1)
    kpe = pos.flatten(2).permute(0, 2, 1)
    queries, qpe = tokens, tokens
    for i in range(2):
        lp = f"{pre}.layers.{i}"
        if i == 0:
            queries = _dec_attn(queries, queries, queries, W, lp + ".self_attn")
        else:
            q = queries + qpe
            queries = queries + _dec_attn(q, q, queries, W, lp + ".self_attn")
        queries = _dln(queries, W, lp + ".norm1")
        q, k = queries + qpe, keys + kpe
        queries = _dln(queries + _dec_attn(q, k, keys, W, lp + ".cross_attn_token_to_image"), W, lp + ".norm2")
        mlp = F.linear(F.relu(F.linear(queries, W[lp + ".mlp.lin1.weight"], W[lp + ".mlp.lin1.bias"])),
                       W[lp + ".mlp.lin2.weight"], W[lp + ".mlp.lin2.bias"])
        queries = _dln(queries + mlp, W, lp + ".norm3")
        q, k = queries + qpe, keys + kpe
        keys = _dln(keys + _dec_attn(k, q, queries, W, lp + ".cross_attn_image_to_token"), W, lp + ".norm4")
    q, k = queries + qpe, keys + kpe
    queries = _dln(queries + _dec_attn(q, k, keys, W, pre + ".final_attn_token_to_image"), W, pre + ".norm_final_attn")
    return queries, keys


def _mlp3(x, W, pre, n=3):
    """mask_decoder.py:158-186."""
    for i in range(n):
        x = F.linear(x, W[f"{pre}.layers.{i}.weight"], W[f"{pre}.layers.{i}.bias"])
        if i < n - 1:
            x = F.relu(x)
    return x


def output_upscaling(src, W, pre="mask_decoder"):
    """mask_decoder.py:53-59: ConvT2x2/s2 -> LayerNorm2d -> GELU -> ConvT2x2/s2 -> GELU.  src [B,256,h,w] -> [B,32,4h,4w]."""
    x = F.conv_transpose2d(src, W[pre + ".output_upscaling.0.weight"], W[pre + ".output_upscaling.0.bias"], stride=2)
    x = F.gelu(_ln2d(x, W, pre + ".output_upscaling.1"))
    x = F.conv_transpose2d(x, W[pre + ".output_upscaling.3.weight"], W[pre + ".output_upscaling.3.bias"], stride=2)
    return F.gelu(x)


def mask_decoder(image_embeddings, image_pe, sparse, dense, W, pre="mask_decoder", return_all=False):
    """mask_decoder.py:71-153 with multimask_output=False -> masks [B,1,4h,4w], iou [B,1]."""
    out_tokens = torch.cat([W[pre + ".iou_token.weight"], W[pre + ".mask_tokens.weight"]], 0)
    tokens = torch.cat((out_tokens.unsqueeze(0).expand(sparse.size(0), -1, -1), sparse), 1)
    src = image_embeddings + dense
    pos = torch.repeat_interleave(image_pe, tokens.shape[0], dim=0)
    b, c, h, w = src.shape
    hs, src2 = two_way_transformer(src, pos, tokens, W, pre + ".transformer")
    iou_tok, mask_toks = hs[:, 0, :], hs[:, 1:5, :]
    up = output_upscaling(src2.transpose(1, 2).view(b, c, h, w), W, pre)
    hyper = torch.stack([_mlp3(mask_toks[:, i, :], W, f"{pre}.output_hypernetworks_mlps.{i}") for i in range(4)], 1)
    b2, c2, h2, w2 = up.shape
    masks = (hyper @ up.view(b2, c2, h2 * w2)).view(b2, -1, h2, w2)
    iou = _mlp3(iou_tok, W, pre + ".iou_prediction_head")
    if return_all:
        return masks, iou, up, hs, src2
    return masks[:, 0:1], iou[:, 0:1]


def init_weights(seed=0, encoder_depth=12, scale=1.0):
    """Seeded random SAM-Med2D weights at the true 256-px geometry with reference key names / shapes
    (build_sam.py:72-121).  Default torch inits are irrelevant for parity; a plain N(0, s) keeps activations O(1)."""
    g = torch.Generator().manual_seed(seed)

    def rn(*shape, s=0.02):
        return torch.randn(*shape, generator=g) * s * scale
    W = {}
    p = "image_encoder"
    W[p + ".patch_embed.proj.weight"] = rn(768, 3, 16, 16, s=0.03)
    W[p + ".patch_embed.proj.bias"] = rn(768)
    W[p + ".pos_embed"] = rn(1, 16, 16, 768)
    for i in range(encoder_depth):
        b = f"{p}.blocks.{i}"
        n = 16 if i in GLOBAL_ATTN else WINDOW
        for nm in ("norm1", "norm2", "Adapter.norm"):
            W[f"{b}.{nm}.weight"] = 1 + rn(768, s=0.1)
            W[f"{b}.{nm}.bias"] = rn(768, s=0.1)
        W[b + ".attn.qkv.weight"] = rn(2304, 768, s=0.04); W[b + ".attn.qkv.bias"] = rn(2304, s=0.1)
        W[b + ".attn.proj.weight"] = rn(768, 768); W[b + ".attn.proj.bias"] = rn(768)
        W[b + ".attn.rel_pos_h"] = rn(2 * n - 1, 64, s=0.2); W[b + ".attn.rel_pos_w"] = rn(2 * n - 1, 64, s=0.2)
        W[b + ".mlp.lin1.weight"] = rn(3072, 768, s=0.04); W[b + ".mlp.lin1.bias"] = rn(3072, s=0.1)
        W[b + ".mlp.lin2.weight"] = rn(768, 3072); W[b + ".mlp.lin2.bias"] = rn(768)
        W[b + ".Adapter.channel.0.weight"] = rn(192, 768, s=0.05); W[b + ".Adapter.channel.2.weight"] = rn(768, 192, s=0.05)
        W[b + ".Adapter.spatial.0.weight"] = rn(768, 768, 3, 3, s=0.01)
        W[b + ".Adapter.spatial.2.weight"] = rn(768, 768, 4, 4, s=0.01)
    W[p + ".neck.0.weight"] = rn(256, 768, 1, 1, s=0.04)
    W[p + ".neck.1.weight"] = 1 + rn(256, s=0.1); W[p + ".neck.1.bias"] = rn(256, s=0.1)
    W[p + ".neck.2.weight"] = rn(256, 256, 3, 3, s=0.03)
    W[p + ".neck.3.weight"] = 1 + rn(256, s=0.1); W[p + ".neck.3.bias"] = rn(256, s=0.1)
    p = "prompt_encoder"
    W[p + ".pe_layer.positional_encoding_gaussian_matrix"] = torch.randn(2, 128, generator=g)
    W[p + ".no_mask_embed.weight"] = rn(1, 256, s=0.5)
    p = "mask_decoder"
    W[p + ".iou_token.weight"] = rn(1, 256, s=1.0)
    W[p + ".mask_tokens.weight"] = rn(4, 256, s=1.0)

    def attn(pre, internal):
        for nm in ("q_proj", "k_proj", "v_proj"):
            W[f"{pre}.{nm}.weight"] = rn(internal, 256, s=0.06); W[f"{pre}.{nm}.bias"] = rn(internal, s=0.05)
        W[pre + ".out_proj.weight"] = rn(256, internal, s=0.06); W[pre + ".out_proj.bias"] = rn(256, s=0.05)
    for i in range(2):
        lp = f"{p}.transformer.layers.{i}"
        attn(lp + ".self_attn", 256)
        attn(lp + ".cross_attn_token_to_image", 128)
        attn(lp + ".cross_attn_image_to_token", 128)
        for k in range(1, 5):
            W[f"{lp}.norm{k}.weight"] = 1 + rn(256, s=0.1); W[f"{lp}.norm{k}.bias"] = rn(256, s=0.1)
        W[lp + ".mlp.lin1.weight"] = rn(2048, 256, s=0.06); W[lp + ".mlp.lin1.bias"] = rn(2048, s=0.05)
        W[lp + ".mlp.lin2.weight"] = rn(256, 2048, s=0.03); W[lp + ".mlp.lin2.bias"] = rn(256, s=0.05)
    attn(p + ".transformer.final_attn_token_to_image", 128)
    W[p + ".transformer.norm_final_attn.weight"] = 1 + rn(256, s=0.1); W[p + ".transformer.norm_final_attn.bias"] = rn(256, s=0.1)
    W[p + ".output_upscaling.0.weight"] = rn(256, 64, 2, 2, s=0.06); W[p + ".output_upscaling.0.bias"] = rn(64, s=0.05)
    W[p + ".output_upscaling.1.weight"] = 1 + rn(64, s=0.1); W[p + ".output_upscaling.1.bias"] = rn(64, s=0.1)
    W[p + ".output_upscaling.3.weight"] = rn(64, 32, 2, 2, s=0.12); W[p + ".output_upscaling.3.bias"] = rn(32, s=0.05)
    for i in range(4):
        hp = f"{p}.output_hypernetworks_mlps.{i}"
        W[hp + ".layers.0.weight"] = rn(256, 256, s=0.06); W[hp + ".layers.0.bias"] = rn(256, s=0.05)
        W[hp + ".layers.1.weight"] = rn(256, 256, s=0.06); W[hp + ".layers.1.bias"] = rn(256, s=0.05)
        W[hp + ".layers.2.weight"] = rn(32, 256, s=0.06); W[hp + ".layers.2.bias"] = rn(32, s=0.05)
    hp = p + ".iou_prediction_head"
    W[hp + ".layers.0.weight"] = rn(256, 256, s=0.06); W[hp + ".layers.0.bias"] = rn(256, s=0.05)
    W[hp + ".layers.1.weight"] = rn(256, 256, s=0.06); W[hp + ".layers.1.bias"] = rn(256, s=0.05)
    W[hp + ".layers.2.weight"] = rn(4, 256, s=0.06); W[hp + ".layers.2.bias"] = rn(4, s=0.05)
    return W
