"""In-context-learning front end of config 5: `TokenCompressor` (576 -> 256 tokens per image) and `MaskTokenEncoder`
(one [H, W] mask -> 64 tokens), model/medplib/model/medplib_arch.py:67-108, over the HIP kernels.

Both are frozen at every stage the reference ships scripts for unless named in `sft_modules` (`mask_encoder`,
`mm_token_compressor` substring matches, train_ds_medplib.py:316-326); this build runs them forward-only in bf16 like the rest of
the frozen trunk.  Weights keep the checkpoint key layout `model.mm_token_compressor.*` / `model.mask_encoder.*`."""
import torch

from .. import ops


def _conv_taps(k, pad):
    return [(ky - pad, kx - pad) for ky in range(k) for kx in range(k)]


class TokenCompressor:
    """x [n, 576, d] -> proj(LayerNorm(AdaptiveAvgPool1d_tokens(x))) [n, num_tokens, d]  (medplib_arch.py:67-77)."""

    def __init__(self, hidden, num_tokens, device, seed=3):
        g = torch.Generator(device=device).manual_seed(seed)
        self.num_tokens, self.hidden = num_tokens, hidden
        self.norm = (torch.ones(hidden, dtype=torch.float32, device=device), torch.zeros(hidden, dtype=torch.float32, device=device))
        self.proj_w = (torch.randn(hidden, hidden, generator=g, device=device) * hidden ** -0.5).to(torch.bfloat16)
        self.proj_b = torch.zeros(hidden, dtype=torch.float32, device=device)

    def load_hf(self, sd, prefix="model.mm_token_compressor."):
        self.norm[0].copy_(sd[prefix + "norm.weight"].float()); self.norm[1].copy_(sd[prefix + "norm.bias"].float())
        self.proj_w.copy_(sd[prefix + "proj.weight"].to(torch.bfloat16)); self.proj_b.copy_(sd[prefix + "proj.bias"].float())

    def export_hf(self, prefix="model.mm_token_compressor."):
        bf = torch.bfloat16
        return {prefix + "norm.weight": self.norm[0].to(bf), prefix + "norm.bias": self.norm[1].to(bf),
                prefix + "proj.weight": self.proj_w, prefix + "proj.bias": self.proj_b.to(bf)}

    def forward(self, feats, n_images, tokens_in):
        """feats [n_images * tokens_in, d] bf16 -> [n_images * num_tokens, d] bf16."""
        pooled = ops.adaptive_avgpool_tokens(feats.view(n_images, tokens_in, self.hidden), self.num_tokens)
        h = ops.layernorm(pooled.view(-1, self.hidden), self.norm[0], self.norm[1], 1e-5)
        return ops.gemm(h, self.proj_w, bias=self.proj_b)


class MaskTokenEncoder:
    """masks [n, 1, H, W] -> LayerNorm(proj(AdaptiveAvgPool1d(flatten(4 x [Conv3x3 s2 p1 + GELU])))) [n, num_tokens, d]
    (medplib_arch.py:80-108).  Layer 1 (one input channel) is a direct kernel; layers 2-4 are NHWC tap-gather im2col + the MFMA
    GEMM with the GELU fused into its epilogue."""
    CH = (64, 128, 256, 256)

    def __init__(self, hidden, num_tokens, device, seed=4):
        g = torch.Generator(device=device).manual_seed(seed)
        self.num_tokens, self.hidden = num_tokens, hidden

        def rn(*shape, s):
            return torch.randn(*shape, generator=g, device=device) * s
        self.w0 = rn(64, 9, s=1 / 3.0)                                   # Conv2d(1, 64): [co, ky*3+kx] fp32
        self.b0 = torch.zeros(64, dtype=torch.float32, device=device)
        self.convs = []
        cin = 64
        for cout in self.CH[1:]:
            # [cout, cin, 3, 3] -> im2col column order (tap-major, channel-minor)
            self.convs.append(((rn(cout, 9 * cin, s=(9 * cin) ** -0.5)).to(torch.bfloat16), torch.zeros(cout, dtype=torch.float32, device=device)))
            cin = cout
        self.proj_w = rn(hidden, 256, s=1 / 16.0).to(torch.bfloat16)
        self.proj_b = torch.zeros(hidden, dtype=torch.float32, device=device)
        self.norm = (torch.ones(hidden, dtype=torch.float32, device=device), torch.zeros(hidden, dtype=torch.float32, device=device))

    def load_hf(self, sd, prefix="model.mask_encoder."):
        self.w0.copy_(sd[prefix + "encoder.0.weight"].float().reshape(64, 9)); self.b0.copy_(sd[prefix + "encoder.0.bias"].float())
        for j, i in enumerate((2, 4, 6)):
            w = sd[prefix + f"encoder.{i}.weight"]                       # [cout, cin, 3, 3]
            self.convs[j][0].copy_(w.permute(0, 2, 3, 1).reshape(w.shape[0], -1).to(torch.bfloat16))
            self.convs[j][1].copy_(sd[prefix + f"encoder.{i}.bias"].float())
        self.proj_w.copy_(sd[prefix + "proj.weight"].to(torch.bfloat16)); self.proj_b.copy_(sd[prefix + "proj.bias"].float())
        self.norm[0].copy_(sd[prefix + "norm.weight"].float()); self.norm[1].copy_(sd[prefix + "norm.bias"].float())

    def export_hf(self, prefix="model.mask_encoder."):
        bf = torch.bfloat16
        sd = {prefix + "encoder.0.weight": self.w0.reshape(64, 1, 3, 3).to(bf), prefix + "encoder.0.bias": self.b0.to(bf)}
        cin = 64
        for j, i in enumerate((2, 4, 6)):
            w, b = self.convs[j]
            sd[prefix + f"encoder.{i}.weight"] = w.view(w.shape[0], 3, 3, cin).permute(0, 3, 1, 2).contiguous()
            sd[prefix + f"encoder.{i}.bias"] = b.to(bf)
            cin = w.shape[0]
        sd.update({prefix + "proj.weight": self.proj_w, prefix + "proj.bias": self.proj_b.to(bf),
                   prefix + "norm.weight": self.norm[0].to(bf), prefix + "norm.bias": self.norm[1].to(bf)})
        return sd

    def forward(self, masks):
        """masks [n, 1, H, W] / [n, H, W] (any float dtype; only channel 0 is used, medplib_arch.py:99-102)
        -> [n * num_tokens, d] bf16."""
        if masks.dim() == 4:
            masks = masks[:, 0]
        if masks.dtype not in (torch.float32, torch.bfloat16):
            masks = masks.float()
        x = ops.conv3x3s2_c1_gelu(masks.contiguous(), self.w0, self.b0)             # [n, H/2, W/2, 64]
        n = x.shape[0]
        for w, b in self.convs:
            H, W, C = x.shape[1:]
            OH, OW = (H + 2 - 3) // 2 + 1, (W + 2 - 3) // 2 + 1
            cols = ops.im2col_nhwc(x, OH, OW, 2, _conv_taps(3, 1))
            x = ops.gemm(cols, w, bias=b, act=ops.ACT_GELU).view(n, OH, OW, w.shape[0])
        toks = ops.adaptive_avgpool_tokens(x.view(n, x.shape[1] * x.shape[2], x.shape[3]), self.num_tokens)
        h = ops.gemm(toks.view(-1, 256), self.proj_w, bias=self.proj_b)
        return ops.layernorm(h, self.norm[0], self.norm[1], 1e-5)
