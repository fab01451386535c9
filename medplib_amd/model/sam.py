"""SAM-Med2D (ViT-B + adapters @256 px, text-prompt encoder, mask decoder) over the HIP kernels.

* `SamImageEncoder` — frozen bf16 trunk (model/segment_anything_med2d/modeling/image_encoder.py:59-238): NHWC token
  layout throughout, convolutions as im2col + MFMA GEMM, windowed/global attention through the fused attention kernel
  with decomposed rel-pos bias.
* `PromptEncoderText` — the text-only slice of PromptEncoder.forward (prompt_encoder.py:140-187): sparse = text embedding,
  dense = no_mask_embed broadcast; dense PE is a model constant computed once (prompt_encoder.py:204-226).
* `MaskDecoder` — trainable fp32 tail with the reference's parameter names (mask_decoder.py:16-153, transformer.py:16-244),
  batched over all prompts (the reference loops one prompt at a time, model/MedPLIB.py:473-502)."""
import os

import numpy as np
import torch
import torch.nn as nn

from .. import ops
from . import autograd_ops as A

GELU = ops.ACT_GELU
_SAM_FUSED = os.environ.get("MP_SAM_FUSED", "1") != "0"                  # A/B: 0 = the frozen encoder as generic launches (forward_generic)
_FUSED_TAIL = os.environ.get("MP_TAIL_FUSED_UPSAMPLER", "1") != "0"     # A/B: 0 = the program and FusedUpsampleMaskFn as two autograd nodes


def _conv_taps(k, pad):
    return [(ky - pad, kx - pad) for ky in range(k) for kx in range(k)]


# ConvTranspose2d(k=4, s=2, p=1): output row oy = 2*iy - 1 + ky.  For output parity py the two contributing (ky, dy=iy-m):
_CONVT_TAPS_1D = {0: [(1, 0), (3, -1)], 1: [(0, 1), (2, 0)]}


def _convt_parity_taps(py, px):
    return [((ky, kx), (dy, dx)) for (ky, dy) in _CONVT_TAPS_1D[py] for (kx, dx) in _CONVT_TAPS_1D[px]]


class SamImageEncoder:
    def __init__(self, cfg, device, seed=2, init_std=0.02):
        self.cfg, self.device = cfg, device
        C, depth, G = cfg.sam_embed_dim, cfg.sam_depth, cfg.sam_grid
        g = torch.Generator(device=device).manual_seed(seed)

        def rn(*shape, std=init_std, dtype=torch.bfloat16):
            return (torch.randn(*shape, generator=g, device=device, dtype=torch.float32) * std).to(dtype)

        def ln(n):
            return [torch.ones(n, dtype=torch.float32, device=device), torch.zeros(n, dtype=torch.float32, device=device)]
        f32 = torch.float32
        self.patch_w, self.patch_b = rn(C, 3 * 16 * 16), rn(C, dtype=f32)
        self.pos = rn(G * G, C)
        self.blocks = []
        for i in range(depth):
            n = G if i in cfg.sam_global_attn else cfg.sam_window
            self.blocks.append({
                "norm1": ln(C), "norm2": ln(C), "ad_norm": ln(C),
                "qkv_w": rn(3 * C, C), "qkv_b": rn(3 * C, dtype=f32), "proj_w": rn(C, C), "proj_b": rn(C, dtype=f32),
                "rph": rn(2 * n - 1, 64, dtype=f32), "rpw": rn(2 * n - 1, 64, dtype=f32),
                "lin1_w": rn(4 * C, C), "lin1_b": rn(4 * C, dtype=f32), "lin2_w": rn(C, 4 * C), "lin2_b": rn(C, dtype=f32),
                "ch0": rn(C // 4, C, dtype=f32), "ch2": rn(C, C // 4, dtype=f32),
                "sp0": rn(C, 9 * C, std=0.01), "sp2": [rn(C, 4 * C, std=0.01) for _ in range(4)]})
            # the transposed conv's four output-parity matrices live stacked (one batched GEMM); "sp2" are views into the stack
            blk = self.blocks[-1]
            blk["sp2_all"] = torch.stack(blk["sp2"]).contiguous()
            blk["sp2"] = [blk["sp2_all"][c] for c in range(4)]
        O = cfg.sam_out_chans
        self.neck0 = rn(O, C); self.neck1 = ln(O); self.neck2 = rn(O, 9 * O); self.neck3 = ln(O)
        self._refresh_derived()

    def _refresh_derived(self):
        """Images of the frozen weights in the layout a kernel wants (the channel gate reads its two matrices transposed)."""
        for b in self.blocks:
            b["ch0T"], b["ch2T"] = b["ch0"].t().contiguous(), b["ch2"].t().contiguous()

    # ------------------------------------------------------------------ reference checkpoint layout (image_encoder.*)
    def load_ref(self, sd, prefix="image_encoder."):
        C = self.cfg.sam_embed_dim

        def put(dst, src):
            dst.copy_(src.to(device=dst.device, dtype=dst.dtype).reshape(dst.shape))
        put(self.patch_w, sd[prefix + "patch_embed.proj.weight"]); put(self.patch_b, sd[prefix + "patch_embed.proj.bias"])
        put(self.pos, sd[prefix + "pos_embed"])
        for i, b in enumerate(self.blocks):
            p = f"{prefix}blocks.{i}."
            for n, k in (("norm1", "norm1"), ("norm2", "norm2"), ("Adapter.norm", "ad_norm")):
                put(b[k][0], sd[p + n + ".weight"]); put(b[k][1], sd[p + n + ".bias"])
            put(b["qkv_w"], sd[p + "attn.qkv.weight"]); put(b["qkv_b"], sd[p + "attn.qkv.bias"])
            put(b["proj_w"], sd[p + "attn.proj.weight"]); put(b["proj_b"], sd[p + "attn.proj.bias"])
            put(b["rph"], sd[p + "attn.rel_pos_h"]); put(b["rpw"], sd[p + "attn.rel_pos_w"])
            put(b["lin1_w"], sd[p + "mlp.lin1.weight"]); put(b["lin1_b"], sd[p + "mlp.lin1.bias"])
            put(b["lin2_w"], sd[p + "mlp.lin2.weight"]); put(b["lin2_b"], sd[p + "mlp.lin2.bias"])
            put(b["ch0"], sd[p + "Adapter.channel.0.weight"]); put(b["ch2"], sd[p + "Adapter.channel.2.weight"])
            w = sd[p + "Adapter.spatial.0.weight"]                          # Conv2d [co, ci, 3, 3] -> [co, (ky,kx,ci)]
            put(b["sp0"], w.permute(0, 2, 3, 1).reshape(C, 9 * C))
            wt = sd[p + "Adapter.spatial.2.weight"]                         # ConvT [ci, co, 4, 4]
            for cls in range(4):
                taps = _convt_parity_taps(cls >> 1, cls & 1)
                packed = torch.stack([wt[:, :, ky, kx].t() for (ky, kx), _ in taps], 1)      # [co, 4, ci]
                put(b["sp2"][cls], packed.reshape(C, 4 * C))
        O = self.cfg.sam_out_chans
        put(self.neck0, sd[prefix + "neck.0.weight"].reshape(O, C))
        put(self.neck1[0], sd[prefix + "neck.1.weight"]); put(self.neck1[1], sd[prefix + "neck.1.bias"])
        put(self.neck2, sd[prefix + "neck.2.weight"].permute(0, 2, 3, 1).reshape(O, 9 * O))
        put(self.neck3[0], sd[prefix + "neck.3.weight"]); put(self.neck3[1], sd[prefix + "neck.3.bias"])
        self._refresh_derived()

    def export_ref(self, prefix="image_encoder."):
        """Inverse of load_ref: the kernel-layout weights back in the SAM-Med2D checkpoint layout (conv kernels un-flattened, the
        four parity classes of the adapter's transposed convolution scattered back into its [ci, co, 4, 4] kernel)."""
        C, O = self.cfg.sam_embed_dim, self.cfg.sam_out_chans
        sd = {prefix + "patch_embed.proj.weight": self.patch_w.reshape(C, 3, 16, 16), prefix + "patch_embed.proj.bias": self.patch_b,
              prefix + "pos_embed": self.pos.reshape(1, self.cfg.sam_grid, self.cfg.sam_grid, C)}
        for i, b in enumerate(self.blocks):
            p = f"{prefix}blocks.{i}."
            for n, k in (("norm1", "norm1"), ("norm2", "norm2"), ("Adapter.norm", "ad_norm")):
                sd[p + n + ".weight"], sd[p + n + ".bias"] = b[k][0], b[k][1]
            sd[p + "attn.qkv.weight"], sd[p + "attn.qkv.bias"] = b["qkv_w"], b["qkv_b"]
            sd[p + "attn.proj.weight"], sd[p + "attn.proj.bias"] = b["proj_w"], b["proj_b"]
            sd[p + "attn.rel_pos_h"], sd[p + "attn.rel_pos_w"] = b["rph"], b["rpw"]
            sd[p + "mlp.lin1.weight"], sd[p + "mlp.lin1.bias"] = b["lin1_w"], b["lin1_b"]
            sd[p + "mlp.lin2.weight"], sd[p + "mlp.lin2.bias"] = b["lin2_w"], b["lin2_b"]
            sd[p + "Adapter.channel.0.weight"], sd[p + "Adapter.channel.2.weight"] = b["ch0"], b["ch2"]
            sd[p + "Adapter.spatial.0.weight"] = b["sp0"].reshape(C, 3, 3, C).permute(0, 3, 1, 2).contiguous()
            wt = torch.zeros((C, C, 4, 4), dtype=b["sp2"][0].dtype, device=b["sp2"][0].device)
            for cls in range(4):
                packed = b["sp2"][cls].reshape(C, 4, C)                           # [co, tap, ci]
                for t, ((ky, kx), _) in enumerate(_convt_parity_taps(cls >> 1, cls & 1)):
                    wt[:, :, ky, kx] = packed[:, t, :].t()
            sd[p + "Adapter.spatial.2.weight"] = wt
        sd[prefix + "neck.0.weight"] = self.neck0.reshape(O, C, 1, 1)
        sd[prefix + "neck.1.weight"], sd[prefix + "neck.1.bias"] = self.neck1
        sd[prefix + "neck.2.weight"] = self.neck2.reshape(O, 3, 3, O).permute(0, 3, 1, 2).contiguous()
        sd[prefix + "neck.3.weight"], sd[prefix + "neck.3.bias"] = self.neck3
        return sd

    # ------------------------------------------------------------------ forward
    def _adapter(self, xn, blk, B):
        """Adapter_Layer.forward on the norm2 output (image_encoder.py:43-56): LN(x + spatial(channel_gate(x) * x))."""
        C, G = self.cfg.sam_embed_dim, self.cfg.sam_grid
        T = G * G
        pooled = ops.token_mean(xn, B, T, C)
        gate = ops.sgemm(ops.sgemm(pooled, blk["ch0"], trans_b=True, act=ops.SACT_RELU), blk["ch2"], trans_b=True,
                         act=ops.SACT_SIGMOID)
        xc = ops.scale_channels(xn, gate, B, T, C)
        half = G // 2
        cols = ops.im2col_nhwc(xc.view(B, G, G, C), half, half, 2, _conv_taps(3, 1))
        s1 = ops.gemm(cols, blk["sp0"], act=ops.ACT_RELU).view(B, half, half, C)
        tmp = torch.empty((B, G, G, C), dtype=torch.bfloat16, device=xn.device)
        # ConvTranspose2d(k 4, s 2, p 1) = four stride-1 GEMMs, one per output parity, each over its own 2 x 2 taps of s1: the same shape
        # four times with four weight matrices -> ONE batched launch (36 launches fewer per step; same-box A/B of the step: 58.75 vs 58.7-59.0 ms,
        # i.e. no measurable difference — these GEMMs are not what the SAM encoder's 3 ms of displacement are made of)
        rows = B * half * half
        cols4 = torch.empty((4, rows, 4 * C), dtype=torch.bfloat16, device=xn.device)
        for cls in range(4):
            ops.im2col_nhwc(s1, half, half, 1, [t for _, t in _convt_parity_taps(cls >> 1, cls & 1)], out=cols4[cls])
        y4 = ops.gemm_batched(cols4, blk["sp2_all"], torch.empty((4, rows, C), dtype=torch.bfloat16, device=xn.device), act=ops.ACT_RELU)
        for cls in range(4):
            ops.scatter_parity(y4[cls], xn, tmp, B, half, half, C, 2, cls >> 1, cls & 1, G, G)
        return ops.layernorm(tmp.view(B * T, C), blk["ad_norm"][0], blk["ad_norm"][1], 1e-5)

    def _fused_ok(self):
        """The round-6 kernels (csrc/sam_encoder.hip) are built for SAM-Med2D's geometry; anything else takes the generic launches."""
        cfg = self.cfg
        return (_SAM_FUSED and cfg.sam_grid == 16 and cfg.sam_window == 14 and cfg.sam_embed_dim == 768 and cfg.sam_embed_dim // cfg.sam_num_heads == 64)

    @ops.with_throughput_tiles
    def forward(self, images):
        """images [B,3,256,256] (f32 or bf16, SAM-normalised) -> image embedding tokens [B, 256, 256] bf16 = the
        reference's [B,256,16,16] in NHWC order (flatten(2).permute(0,2,1), transformer.py:82)."""
        if not self._fused_ok():
            return self.forward_generic(images)
        cfg = self.cfg
        C, G, Hh, ws = cfg.sam_embed_dim, cfg.sam_grid, cfg.sam_num_heads, cfg.sam_window
        B = images.shape[0]
        T, half = G * G, G // 2
        cols = ops.patch_im2col(images.contiguous(), 16, 3 * 16 * 16)
        x = ops.gemm(cols, self.patch_w, bias=self.patch_b)
        blk0 = self.blocks[0]
        x, h = ops.sam_add_layernorm(x, blk0["norm1"][0], blk0["norm1"][1], 1e-6, addend=self.pos)
        nblk = len(self.blocks)
        gate_ev = getattr(self, "gate_events", None)        # set by model_forward for an encoder that runs ahead: block i waits for events[i]
        for i, blk in enumerate(self.blocks):
            if gate_ev is not None and i < len(gate_ev):
                torch.cuda.current_stream().wait_event(gate_ev[i])
            # Block.forward (image_encoder.py:215-236).  The window blocks never materialise the padded windows: qkv / proj run on the map's own
            # 2048 rows, the attention kernel walks the four windows of each map itself (padded keys = the qkv bias, see csrc/sam_encoder.hip)
            qkv = ops.gemm(h, blk["qkv_w"], bias=blk["qkv_b"])
            a = ops.sam_attention(qkv, blk["qkv_b"], blk["rph"], blk["rpw"], B, Hh, G, 0 if i in cfg.sam_global_attn else ws)
            x = ops.gemm(a, blk["proj_w"], bias=blk["proj_b"], residual=x)
            xn, part = ops.sam_layernorm_colsum(x, blk["norm2"][0], blk["norm2"][1], 1e-6)
            mlp = ops.gemm(ops.gemm(xn, blk["lin1_w"], bias=blk["lin1_b"], act=GELU), blk["lin2_w"], bias=blk["lin2_b"])
            # Adapter_Layer.forward (image_encoder.py:43-56)
            gate = ops.sam_channel_gate(part, B, T, blk["ch0T"], blk["ch2T"])
            s1 = ops.gemm(ops.sam_im2col_scaled(xn, gate, B, G, C), blk["sp0"], act=ops.ACT_RELU)
            cols4 = ops.sam_im2col_parity4(s1, B, half, C)
            y4 = ops.gemm_batched(cols4, blk["sp2_all"], torch.empty((4, B * half * half, C), dtype=torch.bfloat16, device=x.device), act=ops.ACT_RELU)
            nxt = self.blocks[i + 1]["norm1"] if i + 1 < nblk else (None, None)
            x, h = ops.sam_block_tail(y4, xn, x, mlp, blk["ad_norm"][0], blk["ad_norm"][1], 1e-5, nxt[0], nxt[1], 1e-6, B, G)
        return self._neck(x, B)

    def _neck(self, x, B):
        cfg = self.cfg
        G, O = cfg.sam_grid, cfg.sam_out_chans
        y = ops.gemm(x, self.neck0)
        y = ops.layernorm(y, self.neck1[0], self.neck1[1], 1e-6)
        cols = ops.im2col_nhwc(y.view(B, G, G, O), G, G, 1, _conv_taps(3, 1))
        y = ops.gemm(cols, self.neck2)
        y = ops.layernorm(y, self.neck3[0], self.neck3[1], 1e-6)
        return y.view(B, G * G, O)

    @ops.with_throughput_tiles
    def forward_generic(self, images):
        """The same encoder as generic launches (padded windows through the general attention kernel, a rel-pos table kernel, one launch per
        elementwise step): any geometry; also the A/B reference of the fused path (MP_SAM_FUSED=0)."""
        cfg = self.cfg
        C, G, Hh, ws = cfg.sam_embed_dim, cfg.sam_grid, cfg.sam_num_heads, cfg.sam_window
        B = images.shape[0]
        T = G * G
        cols = ops.patch_im2col(images.contiguous(), 16, 3 * 16 * 16)
        x = ops.gemm(cols, self.patch_w, bias=self.patch_b)
        x = ops.add_rows(x, self.pos)
        for i, blk in enumerate(self.blocks):
            h = ops.layernorm(x, blk["norm1"][0], blk["norm1"][1], 1e-6)
            if i in cfg.sam_global_attn:
                Bw, hh = B, G
                hw = h
            else:
                hw = ops.window_partition(h.view(B, G, G, C), ws)
                Bw, hh = hw.shape[0], ws
            S = hh * hh
            qkv = ops.gemm(hw.view(Bw * S, C), blk["qkv_w"], bias=blk["qkv_b"])
            rel_h, rel_w = ops.relpos_tables(qkv, blk["rph"], blk["rpw"], Bw, Hh, hh, hh)
            q5 = qkv.view(Bw, S, 3, Hh, C // Hh)
            a = ops.attention(q5[:, :, 0], q5[:, :, 1], q5[:, :, 2], rel_h=rel_h, rel_w=rel_w)
            if i in cfg.sam_global_attn:
                x = ops.gemm(a.view(B * T, C), blk["proj_w"], bias=blk["proj_b"], residual=x)
            else:
                p = ops.gemm(a.view(Bw * S, C), blk["proj_w"], bias=blk["proj_b"])
                x = ops.window_unpartition_add(p, x.view(B, G, G, C), ws).view(B * T, C)
            xn = ops.layernorm(x, blk["norm2"][0], blk["norm2"][1], 1e-6)
            mlp = ops.gemm(ops.gemm(xn, blk["lin1_w"], bias=blk["lin1_b"], act=GELU), blk["lin2_w"], bias=blk["lin2_b"])
            ad = self._adapter(xn, blk, B)
            x = ops.add3(x, mlp, ad)
        return self._neck(x, B)


class PromptEncoderText(nn.Module):
    """Frozen; holds only what the text-prompt path touches."""

    def __init__(self, embed_dim=256, grid=16):
        super().__init__()
        self.grid = grid
        self.no_mask_embed = nn.Embedding(1, embed_dim)
        self.pe_layer = nn.Module()
        self.pe_layer.register_buffer("positional_encoding_gaussian_matrix", torch.randn(2, embed_dim // 2))
        self._pe_cache = None
        for p in self.parameters():
            p.requires_grad = False

    def dense_pe_tokens(self):
        """get_dense_pe() (prompt_encoder.py:62-71) as tokens [h*w, C] fp32 on the device; a model constant, so it is
        evaluated once on the host exactly as the reference does (fp32 cumsum grid, sin/cos) and cached."""
        G = self.pe_layer.positional_encoding_gaussian_matrix
        if self._pe_cache is None or self._pe_cache.device != G.device:
            h = w = self.grid
            grid = torch.ones((h, w), dtype=torch.float32)
            y = (grid.cumsum(0) - 0.5) / h
            x = (grid.cumsum(1) - 0.5) / w
            c = 2 * torch.stack([x, y], -1) - 1
            c = 2 * np.pi * (c @ G.detach().float().cpu())
            pe = torch.cat([torch.sin(c), torch.cos(c)], -1)              # [h,w,C] == permute(2,0,1) in NHWC order
            self._pe_cache = pe.reshape(h * w, -1).contiguous().to(G.device)
        return self._pe_cache


class _Attention(nn.Module):
    def __init__(self, dim, heads, downsample=1):
        super().__init__()
        inner = dim // downsample
        self.num_heads = heads
        self.q_proj, self.k_proj = nn.Linear(dim, inner), nn.Linear(dim, inner)
        self.v_proj, self.out_proj = nn.Linear(dim, inner), nn.Linear(inner, dim)

    def forward(self, q, k, v):
        q = A.linear(q, self.q_proj.weight, self.q_proj.bias)
        k = A.linear(k, self.k_proj.weight, self.k_proj.bias)
        v = A.linear(v, self.v_proj.weight, self.v_proj.bias)
        o = A.AttentionCoreFn.apply(q, k, v, self.num_heads)
        return A.linear(o, self.out_proj.weight, self.out_proj.bias)


class _MLPBlock(nn.Module):
    def __init__(self, dim, mlp_dim):
        super().__init__()
        self.lin1, self.lin2 = nn.Linear(dim, mlp_dim), nn.Linear(mlp_dim, dim)

    def forward(self, x):
        return A.linear(A.linear(x, self.lin1.weight, self.lin1.bias, ops.SACT_RELU), self.lin2.weight, self.lin2.bias)


class _TwoWayBlock(nn.Module):
    def __init__(self, dim, heads, mlp_dim, skip_first_layer_pe):
        super().__init__()
        self.self_attn = _Attention(dim, heads)
        self.norm1 = nn.LayerNorm(dim)
        self.cross_attn_token_to_image = _Attention(dim, heads, 2)
        self.norm2 = nn.LayerNorm(dim)
        self.mlp = _MLPBlock(dim, mlp_dim)
        self.norm3 = nn.LayerNorm(dim)
        self.norm4 = nn.LayerNorm(dim)
        self.cross_attn_image_to_token = _Attention(dim, heads, 2)
        self.skip_first_layer_pe = skip_first_layer_pe

    @staticmethod
    def _ln(m, x):
        return A.layernorm(x, m.weight, m.bias, m.eps)

    def forward(self, queries, keys, query_pe, key_pe):
        if self.skip_first_layer_pe:
            queries = self.self_attn(queries, queries, queries)
        else:
            q = A.add(queries, query_pe)
            queries = A.add(queries, self.self_attn(q, q, queries))
        queries = self._ln(self.norm1, queries)
        q, k = A.add(queries, query_pe), A.add(keys, key_pe)
        queries = self._ln(self.norm2, A.add(queries, self.cross_attn_token_to_image(q, k, keys)))
        queries = self._ln(self.norm3, A.add(queries, self.mlp(queries)))
        q, k = A.add(queries, query_pe), A.add(keys, key_pe)
        keys = self._ln(self.norm4, A.add(keys, self.cross_attn_image_to_token(k, q, queries)))
        return queries, keys


class _TwoWayTransformer(nn.Module):
    def __init__(self, depth=2, dim=256, heads=8, mlp_dim=2048):
        super().__init__()
        self.layers = nn.ModuleList([_TwoWayBlock(dim, heads, mlp_dim, i == 0) for i in range(depth)])
        self.final_attn_token_to_image = _Attention(dim, heads, 2)
        self.norm_final_attn = nn.LayerNorm(dim)

    def forward(self, keys, key_pe, tokens):
        queries = tokens
        for layer in self.layers:
            queries, keys = layer(queries, keys, tokens, key_pe)
        q, k = A.add(queries, tokens), A.add(keys, key_pe)
        queries = A.add(queries, self.final_attn_token_to_image(q, k, keys))
        queries = A.layernorm(queries, self.norm_final_attn.weight, self.norm_final_attn.bias, self.norm_final_attn.eps)
        return queries, keys


class _MLP(nn.Module):
    def __init__(self, i, h, o, n):
        super().__init__()
        dims = [i] + [h] * (n - 1) + [o]
        self.layers = nn.ModuleList(nn.Linear(a, b) for a, b in zip(dims[:-1], dims[1:]))

    def forward(self, x):
        n = len(self.layers)
        for i, l in enumerate(self.layers):
            x = A.linear(x, l.weight, l.bias, ops.SACT_RELU if i < n - 1 else ops.SACT_NONE)
        return x


class _LayerNorm2d(nn.Module):
    def __init__(self, c, eps=1e-6):
        super().__init__()
        self.weight, self.bias, self.eps = nn.Parameter(torch.ones(c)), nn.Parameter(torch.zeros(c)), eps


class MaskDecoder(nn.Module):
    """Parameter names and shapes identical to the reference's MaskDecoder(transformer_dim=256, 3 multimask outputs)."""

    def __init__(self, dim=256, grid=16):
        super().__init__()
        self.dim, self.grid = dim, grid
        self.transformer = _TwoWayTransformer(2, dim, 8, 2048)
        self.iou_token = nn.Embedding(1, dim)
        self.mask_tokens = nn.Embedding(4, dim)
        self.output_upscaling = nn.Sequential(nn.ConvTranspose2d(dim, dim // 4, 2, 2), _LayerNorm2d(dim // 4), nn.GELU(),
                                              nn.ConvTranspose2d(dim // 4, dim // 8, 2, 2), nn.GELU())
        self.output_hypernetworks_mlps = nn.ModuleList([_MLP(dim, dim, dim // 8, 3) for _ in range(4)])
        self.iou_prediction_head = _MLP(dim, 256, 4, 3)
        self.fused_bf16_upsampler = False       # opt-in inference path (config.fused_bf16_upsampler); packed weights cached
        self._packed = None
        # text_hidden_fcs + two-way transformer + heads as one launch each way (tail_program.py); MP_TAIL_PROGRAM=0: the op-by-op path (A/B)
        self.use_program = os.environ.get("MP_TAIL_PROGRAM", "1") != "0"
        self._runner, self._runner_key = None, None
        # workgroups of the program's persistent grid: 256 (one per CU) when the tail is on the step's critical path (decoder adapters training);
        # the model sets 64 while the tail runs hidden on its own stream beside the next step's decoder — resident workgroups that wait at a
        # barrier still hold their CU against the decoder's GEMM tiles (measured round 5, headline step: 61.45 ms at 256, 59.58 at 64, 59.4-59.5
        # with the op-by-op tail; LoRA step: 131.6 at 256 vs 131.7-131.8).  MP_TAIL_GRID overrides both.
        self.program_grid = 256

    def train(self, mode=True):
        self._packed = None                     # weights may change while training: re-pack on the next inference call
        return super().train(mode)

    def _runner_for(self, dense_pe_tokens, no_mask_embed, fcs, fused=False):
        """The program runner bound to these constants / this text_hidden_fcs pair (tail_program.TailRunner), built on first use."""
        from ..tail_program import TailRunner
        key = (dense_pe_tokens.data_ptr(), no_mask_embed.data_ptr(), None if fcs is None else (id(fcs[0]), id(fcs[1])), bool(fused))
        if self._runner is None or self._runner_key != key:
            self._runner, self._runner_key = TailRunner(self, dense_pe_tokens, no_mask_embed, fcs=fcs, fused_upsampler=fused), key
        if "MP_TAIL_GRID" not in os.environ:
            self._runner.grid = int(self.program_grid)
        return self._runner

    def forward(self, image_tokens, dense_pe_tokens, no_mask_embed, text_embeds, fcs=None, hidden_rows=None):
        """image_tokens [n, h*w, C] fp32 (NHWC order), dense_pe_tokens [h*w, C], no_mask_embed [1, C], text_embeds [n,1,C]
        -> low-res mask logits [n, 4h, 4w] (mask token 0, multimask_output=False) and iou prediction [n].
        fcs = (fc1, fc2) + hidden_rows [n, hidden]: text_hidden_fcs runs in front (text_embeds is then ignored) — the form the training
        step uses, so that the whole trainable tail up to the upsampler is ONE launch each way (tail_program.py)."""
        n, T, C = image_tokens.shape
        g = self.grid
        if self.use_program and n > 0 and T == g * g and C == self.dim:
            x_in = hidden_rows if fcs is not None else text_embeds.reshape(n, C)
            if self.fused_bf16_upsampler and g % 16 == 0 and torch.is_grad_enabled() and C == 256 and _FUSED_TAIL:
                # training through the fused bf16 upsampler: program + upsampler as ONE autograd node (tail_program.TailFusedFn)
                return self._runner_for(dense_pe_tokens, no_mask_embed, fcs, fused=True)(x_in.float(), image_tokens)
            src, hyper0, iou = self._runner_for(dense_pe_tokens, no_mask_embed, fcs)(x_in.float(), image_tokens)
            return self._upscale(src, hyper0, n), iou
        if fcs is not None:
            text_embeds = A.linear(A.linear(hidden_rows, fcs[0].weight, fcs[0].bias, ops.SACT_RELU), fcs[1].weight, fcs[1].bias).view(n, 1, -1)
        tokens = A.BuildTokensFn.apply(self.iou_token.weight, self.mask_tokens.weight, text_embeds)
        src = A.add(image_tokens, no_mask_embed.view(-1))                 # src = image_embeddings + dense (broadcast over tokens)
        hs, src = self.transformer(src, dense_pe_tokens, tokens)
        iou_tok, mask_tok0 = hs[:, 0, :], hs[:, 1, :]
        hyper0 = self.output_hypernetworks_mlps[0](mask_tok0)              # [n, 32]; mask slice 0 (mask_decoder.py:102-108)
        return self._upscale(src, hyper0, n), self.iou_prediction_head(iou_tok)[:, 0]

    def _upscale(self, src, hyper0, n):
        """output_upscaling + hyper_in @ upscaled_embedding (mask_decoder.py:53-59,141-148): src [n, h*w, C], hyper0 [n, C/8] -> [n, 4h, 4w]."""
        g, C = self.grid, self.dim
        if self.fused_bf16_upsampler and not torch.is_grad_enabled() and g % 16 == 0:
            # inference: ConvT -> LayerNorm2d -> GELU -> ConvT -> GELU -> hypernetwork product in ONE bf16 pass over HBM
            # (mp_mask_upsample_fused_bf16), the arithmetic the reference itself runs under `--precision bf16`
            if self._packed is None:
                ct1, ln, ct2 = self.output_upscaling[0], self.output_upscaling[1], self.output_upscaling[3]
                w1p, w2p = ops.pack_upsampler_weights(ct1.weight.detach(), ct2.weight.detach())
                self._packed = (w1p, ct1.bias.detach().float().contiguous(), ln.weight.detach().float().contiguous(),
                                ln.bias.detach().float().contiguous(), w2p, ct2.bias.detach().float().contiguous(), float(ln.eps))
            w1p, b1, lw, lb, w2p, b2, eps = self._packed
            _, masks = ops.mask_upsample_fused(ops.cast_to_bf16(src.contiguous()).view(n, g * g, C), w1p, b1, lw, lb, w2p, b2, g, g,
                                               hyper=hyper0.contiguous(), want_up=False, eps=eps)
            return masks
        if self.fused_bf16_upsampler and g % 16 == 0:
            # training: the same kernel, differentiated by one recomputing backward kernel (A.FusedUpsampleMaskFn)
            ct1, ln, ct2 = self.output_upscaling[0], self.output_upscaling[1], self.output_upscaling[3]
            return A.FusedUpsampleMaskFn.apply(src, ct1.weight, ct1.bias, ln.weight, ln.bias, ct2.weight, ct2.bias, hyper0, g, float(ln.eps))
        up = A.ConvT2x2Fn.apply(src.view(n, g, g, C), self.output_upscaling[0].weight, self.output_upscaling[0].bias)
        ln = self.output_upscaling[1]
        up = A.GeluFn.apply(A.layernorm(up, ln.weight, ln.bias, ln.eps))
        up = A.GeluFn.apply(A.ConvT2x2Fn.apply(up, self.output_upscaling[3].weight, self.output_upscaling[3].bias))
        masks = A.HyperDotFn.apply(hyper0, up.view(n, 16 * g * g, C // 8))
        return masks.view(n, 4 * g, 4 * g)
