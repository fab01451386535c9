"""GPU parity of the SAM-Med2D encoder's own kernels (csrc/sam_encoder.hip, round 6) through the C ABI.

Three kinds of check:
* against plain fp32 torch restatements of the reference's arithmetic (window_partition with zero padding -> Attention.forward with
  add_decomposed_rel_pos -> window_unpartition, image_encoder.py:217-230, 280-296, 299-421; Adapter_Layer, :43-56): bf16 tolerances stated;
* against the generic launches they replace (same rounding points): bit-identical where the arithmetic order is the same (im2col forms, the
  block tail, norm2), 1-2 bf16 ulps where a reduction order changed (single-pass softmax, slab sums);
* the whole encoder fused vs generic at batch 2 and 8, and the reference-module golden through the fused path (tests/test_gpu_model.py:186 runs
  the default = fused path; here both are compared on the same weights)."""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

C, H, G, WS = 768, 12, 16, 14


def _stat(name, got, ref, atol, rtol=0.0):
    got, ref = got.float().cpu(), ref.float().cpu()
    err = (got - ref).abs()
    bad = err > atol + rtol * ref.abs()
    msg = f"{name}: max|err|={err.max().item():.3e} mean|err|={err.mean().item():.3e} ref absmax={ref.abs().max().item():.3e} bad={int(bad.sum())}/{bad.numel()}"
    print(msg)
    assert not bad.any(), msg


def _ref_attention(qkv_in, w, bias, rph, rpw, B, window):
    """fp32 restatement: x [B, G, G, C] (the norm1 output, bf16 values) -> Attention with padded windows -> [B, G, G, C] BEFORE proj.
    qkv = bf16(x W^T + b) per token like the GEMM; padded tokens are zero rows (their qkv = bf16(b))."""
    x = qkv_in.float().view(B, G, G, C)
    n = window if window else G
    if window:
        pad = (n - G % n) % n
        xp = F.pad(x, (0, 0, 0, pad, 0, pad))
        Gp = G + pad
        nw = Gp // n
        win = xp.view(B, nw, n, nw, n, C).permute(0, 1, 3, 2, 4, 5).reshape(-1, n, n, C)
    else:
        win, nw, Gp = x, 1, G
    Bw = win.shape[0]
    qkv = (win.reshape(-1, C) @ w.float().t() + bias.float()).to(torch.bfloat16).float().view(Bw, n * n, 3, H, 64).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0], qkv[1], qkv[2]                                           # [Bw, H, S, 64]
    attn = (q * 64 ** -0.5) @ k.transpose(-2, -1)
    idx = torch.arange(n)[:, None] - torch.arange(n)[None, :] + (n - 1)
    Rh, Rw = rph.float()[idx], rpw.float()[idx]                                 # [n, n, 64]
    rq = q.reshape(Bw, H, n, n, 64)
    rel_h = torch.einsum("bhywc,ykc->bhywk", rq, Rh)
    rel_w = torch.einsum("bhywc,wkc->bhywk", rq, Rw)
    attn = (attn.view(Bw, H, n, n, n, n) + rel_h[..., :, None] + rel_w[..., None, :]).view(Bw, H, n * n, n * n)
    o = (attn.softmax(-1) @ v).view(Bw, H, n, n, 64).permute(0, 2, 3, 1, 4).reshape(Bw, n, n, C)
    if window:
        o = o.view(B, nw, nw, n, n, C).permute(0, 1, 3, 2, 4, 5).reshape(B, Gp, Gp, C)[:, :G, :G]
    return o.reshape(B * G * G, C)


@pytest.mark.parametrize("window", [WS, 0])
def test_sam_attention_vs_fp32_restatement_and_generic_launches(dev, window):
    from medplib_amd import ops
    g = torch.Generator().manual_seed(11 + window)
    B = 3
    n = window if window else G
    h = torch.randn(B * G * G, C, generator=g).to(torch.bfloat16)
    w = (torch.randn(3 * C, C, generator=g) * 0.04).to(torch.bfloat16)
    bias = torch.randn(3 * C, generator=g) * 0.5                               # a LARGE bias: the padded keys must matter
    rph, rpw = torch.randn(2 * n - 1, 64, generator=g) * 0.3, torch.randn(2 * n - 1, 64, generator=g) * 0.3
    hd, wd, bd, rhd, rwd = h.to(dev), w.to(dev), bias.to(dev), rph.to(dev), rpw.to(dev)
    qkv = ops.gemm(hd, wd, bias=bd)
    out = ops.sam_attention(qkv, bd, rhd, rwd, B, H, G, window)
    ref = _ref_attention(h, w, bias, rph, rpw, B, window)
    # outputs are convex combinations of v rows (|v| <~ 3): P is rounded to bf16 (2^-9 relative per weight), the output once more
    _stat(f"sam_attention window={window} vs fp32 restatement", out, ref, atol=0.03)
    assert (out.float().cpu() - ref).abs().mean().item() < 3e-3
    # the launches it replaces, same qkv weights: partition (zero pad) -> qkv GEMM -> tables -> general attention -> crop
    if window:
        hw = ops.window_partition(hd.view(B, G, G, C), window)
        Bw = hw.shape[0]
    else:
        hw, Bw = hd, B
    S = n * n
    qkv_w = ops.gemm(hw.view(Bw * S, C), wd, bias=bd)
    rel_h, rel_w = ops.relpos_tables(qkv_w, rhd, rwd, Bw, H, n, n)
    q5 = qkv_w.view(Bw, S, 3, H, 64)
    a = ops.attention(q5[:, :, 0], q5[:, :, 1], q5[:, :, 2], rel_h=rel_h, rel_w=rel_w).reshape(Bw, n, n, C)
    if window:
        nw = 2
        a = a.view(B, nw, nw, n, n, C).permute(0, 1, 3, 2, 4, 5).reshape(B, nw * n, nw * n, C)[:, :G, :G]
    # same operands, same rel-pos arithmetic; single-pass instead of online softmax: a few bf16 ulps of the output
    _stat(f"sam_attention window={window} vs generic launches", out, a.reshape(B * G * G, C), atol=0.02)


def test_row_kernels_match_the_launches_they_replace(dev):
    from medplib_amd import ops
    g = torch.Generator().manual_seed(5)
    B = 8
    T = G * G
    x = torch.randn(B * T, C, generator=g).to(torch.bfloat16).to(dev)
    pos = torch.randn(T, C, generator=g).to(torch.bfloat16).to(dev)
    w1, b1 = (1 + 0.1 * torch.randn(C, generator=g)).to(dev), (0.1 * torch.randn(C, generator=g)).to(dev)
    w2, b2 = (1 + 0.1 * torch.randn(C, generator=g)).to(dev), (0.1 * torch.randn(C, generator=g)).to(dev)
    # pos add + norm1
    xs, hn = ops.sam_add_layernorm(x, w1, b1, 1e-6, addend=pos)
    xs_ref = ops.add_rows(x, pos)
    assert torch.equal(xs, xs_ref) and torch.equal(hn, ops.layernorm(xs_ref, w1, b1, 1e-6))
    assert torch.equal(ops.sam_add_layernorm(x, w1, b1, 1e-6), ops.layernorm(x, w1, b1, 1e-6))
    # norm2 + slab sums
    xn, part = ops.sam_layernorm_colsum(x, w2, b2, 1e-6)
    xn_ref = ops.layernorm(x, w2, b2, 1e-6)
    assert torch.equal(xn, xn_ref)
    mean_ref = xn_ref.float().view(B, T, C).mean(1)
    _stat("slab sums -> mean", part.view(B, T // 16, C).sum(1) / T, mean_ref, atol=1e-5, rtol=1e-5)
    # channel gate
    ch0, ch2 = (torch.randn(C // 4, C, generator=g) * 0.05).to(dev), (torch.randn(C, C // 4, generator=g) * 0.05).to(dev)
    gate = ops.sam_channel_gate(part, B, T, ch0.t().contiguous(), ch2.t().contiguous())
    gate_ref = torch.sigmoid(torch.relu(mean_ref @ ch0.t()) @ ch2.t())
    _stat("channel gate", gate, gate_ref, atol=1e-5, rtol=1e-5)
    gate_old = ops.sgemm(ops.sgemm(ops.token_mean(xn_ref, B, T, C), ch0, trans_b=True, act=ops.SACT_RELU), ch2, trans_b=True, act=ops.SACT_SIGMOID)
    _stat("channel gate vs the three launches", gate, gate_old, atol=1e-5, rtol=1e-5)
    # im2col of gate * x
    taps3 = [(ky - 1, kx - 1) for ky in range(3) for kx in range(3)]
    cols = ops.sam_im2col_scaled(xn, gate_old, B, G, C)
    cols_ref = ops.im2col_nhwc(ops.scale_channels(xn_ref, gate_old, B, T, C).view(B, G, G, C), G // 2, G // 2, 2, taps3)
    assert torch.equal(cols, cols_ref)
    # the four parity gathers
    from medplib_amd.model.sam import _convt_parity_taps
    half = G // 2
    s1 = torch.randn(B, half, half, C, generator=g).to(torch.bfloat16).to(dev)
    cols4 = ops.sam_im2col_parity4(s1, B, half, C)
    for cls in range(4):
        ref = ops.im2col_nhwc(s1, half, half, 1, [t for _, t in _convt_parity_taps(cls >> 1, cls & 1)])
        assert torch.equal(cols4[cls], ref), cls
    # the block tail
    y4 = torch.randn(4, B * half * half, C, generator=g).to(torch.bfloat16).to(dev)
    mlp = torch.randn(B * T, C, generator=g).to(torch.bfloat16).to(dev)
    aw, ab = (1 + 0.1 * torch.randn(C, generator=g)).to(dev), (0.1 * torch.randn(C, generator=g)).to(dev)
    xo, ho = ops.sam_block_tail(y4, xn, x, mlp, aw, ab, 1e-5, w1, b1, 1e-6, B, G)
    tmp = torch.empty((B, G, G, C), dtype=torch.bfloat16, device=dev)
    for cls in range(4):
        ops.scatter_parity(y4[cls], xn, tmp, B, half, half, C, 2, cls >> 1, cls & 1, G, G)
    ad = ops.layernorm(tmp.view(B * T, C), aw, ab, 1e-5)
    xo_ref = ops.add3(x, mlp, ad)
    assert torch.equal(xo, xo_ref) and torch.equal(ho, ops.layernorm(xo_ref, w1, b1, 1e-6))
    xo2, ho2 = ops.sam_block_tail(y4, xn, x, mlp, aw, ab, 1e-5, None, None, 1e-6, B, G)
    assert torch.equal(xo2, xo_ref) and ho2 is None


@pytest.mark.parametrize("B", [2, 8])
def test_encoder_fused_vs_generic(dev, B):
    from medplib_amd.model.config import MedPLIBConfig
    from medplib_amd.model.sam import SamImageEncoder
    enc = SamImageEncoder(MedPLIBConfig.tiny(), dev, seed=3)
    img = torch.randn(B, 3, 256, 256, generator=torch.Generator().manual_seed(B)).to(dev)
    assert enc._fused_ok()
    a = enc.forward(img)
    b = enc.forward_generic(img)
    # the output is LayerNorm2d-normalised (O(1)); the two paths differ by softmax / mean reduction orders only, amplified through 12 bf16 blocks
    _stat(f"SAM encoder fused vs generic, B={B}", a, b, atol=0.06)
    assert (a.float() - b.float()).abs().mean().item() < 8e-3
    assert torch.equal(a, enc.forward(img))                                      # reproducible


def test_encoder_fused_and_generic_against_reference_golden(dev, golden_dir):
    import os
    from oracle import sam as OS
    from medplib_amd.model.config import MedPLIBConfig
    from medplib_amd.model.sam import SamImageEncoder
    gld = np.load(os.path.join(golden_dir, "sam_reference.npz"))
    enc = SamImageEncoder(MedPLIBConfig.tiny(), dev)
    enc.load_ref(OS.init_weights(seed=int(gld["weight_seed"])), "image_encoder.")
    img = torch.from_numpy(gld["image"]).to(dev)
    ref = torch.from_numpy(gld["image_embedding"])[0].permute(1, 2, 0).reshape(256, 256)
    for name, out in (("fused", enc.forward(img)), ("generic", enc.forward_generic(img))):
        _stat(f"sam image embedding ({name}) vs REFERENCE golden", out[0], ref, atol=0.08)
        assert (out[0].float().cpu() - ref).abs().mean().item() < 0.01


@pytest.mark.parametrize("M,N,K,act", [(2048, 3072, 768, "gelu"), (512, 768, 6912, "relu"), (2048, 768, 768, "relu")])
def test_tower_policy_gemms_on_320_row_tiles_with_relu_and_gelu(dev, M, N, K, act):
    """Round 6: under the frozen towers' whole-tile policy the SAM encoder's mlp.lin1 (erf-GELU) and Adapter.spatial (ReLU, M = 512) GEMMs take
    the 320-row kernel's EPI_GELU / EPI_RELU families instead of the 128 x 128 kernel.  Against an fp32 product of the same bf16 operands
    (bf16 output: 2^-8 relative) and against the 128 x 128 kernel's result for the same call (same activation expression on a differently
    ordered fp32 sum: a bf16 ulp)."""
    from medplib_amd import ops
    g = torch.Generator().manual_seed(M + N)
    a = (torch.randn(M, K, generator=g) * 0.5).to(torch.bfloat16)
    w = (torch.randn(N, K, generator=g) * K ** -0.5).to(torch.bfloat16)
    bias = torch.randn(N, generator=g) * 0.2
    code = ops.ACT_GELU if act == "gelu" else ops.ACT_RELU
    ad, wd, bd = a.to(dev), w.to(dev), bias.to(dev)
    with ops.throughput_tiles():
        y = ops.gemm(ad, wd, bias=bd, act=code)
        assert ops.gemm_last_kernel() == 320
    ops.gemm_tile_policy(0)
    try:
        y0 = ops.gemm(ad, wd, bias=bd, act=code)
        assert ops.gemm_last_kernel() != 320
    finally:
        ops.gemm_tile_policy(-1)
    pre = a.float() @ w.float().t() + bias
    ref = F.gelu(pre) if act == "gelu" else torch.relu(pre)
    _stat(f"320-row {act} M={M} N={N} K={K} vs fp32", y, ref, atol=2e-2, rtol=1e-2)
    _stat(f"320-row {act} vs the other kernel", y, y0, atol=1.6e-2, rtol=8e-3)
