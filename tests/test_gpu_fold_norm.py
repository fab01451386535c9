"""config.fold_input_norm (round 6): both RMSNorms of a frozen top-1 MoE layer folded into their consumer GEMMs (rstd in the epilogue of the qkv + RoPE
and of the experts' gate|up GEMM, the norm weight in the frozen weight's columns) against the unfolded kernels on the same weights, at the 7B
dimensions the folded entry points are built for (320-row-kernel shapes).  The fold moves HF's rounding points (the normalised row is never a bf16
tensor, the folded weight is rounded instead), so the two paths agree to bf16 noise, not bit for bit; the MoE gate is computed from HF's bf16 h in
both (from the stream behind the attention, which carries the fold's noise of the qkv projection).  The full-depth check against the oracle is tests/test_gpu_model.py::
test_full_depth_parity_batch8_folded_norms."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_folded_consumer_gemms_against_the_unfolded_kernels(dev):
    from medplib_amd import ops
    g = torch.Generator().manual_seed(3)
    T, d, H, D, S = 1536, 4096, 32, 128, 512
    x = (torch.randn(T, d, generator=g) * 1.3).to(torch.bfloat16).to(dev)
    ln = (1 + 0.3 * torch.randn(d, generator=g)).to(dev)
    w = (torch.randn(3 * d, d, generator=g) * d ** -0.5).to(torch.bfloat16).to(dev)
    wr = ops.rope_interleave_qkv(w, H, D)
    wf = (wr.float() * ln[None, :]).to(torch.bfloat16)
    pos = torch.arange(S, dtype=torch.float32)
    inv = 1.0 / (10000 ** (torch.arange(0, D, 2, dtype=torch.float32) / D))
    cos, sin = torch.outer(pos, inv).cos().contiguous().to(dev), torch.outer(pos, inv).sin().contiguous().to(dev)
    h = ops.rmsnorm(x, ln, 1e-6)
    rstd, _, _ = ops.rmsnorm_gate_rstd(x, ln, 1e-6)
    assert float((rstd - torch.rsqrt(x.float().pow(2).mean(1) + 1e-6)).abs().max()) < 1e-6
    ref = ops.gemm_qkv_rope(h, wr, cos, sin, S, H, D)
    out = ops.gemm_qkv_rope(x, wf, cos, sin, S, H, D, row_scale=rstd)
    assert ops.gemm_last_kernel() == 320
    err = (out.float() - ref.float()).abs()
    print(f"qkv + RoPE folded vs unfolded: max {err.max().item():.3e} mean {err.mean().item():.3e} ref absmax {ref.float().abs().max().item():.3e}")
    # two bf16 roundings of the row against one of the weight, both summed over K = 4096 in fp32: a few bf16 ulps of the output
    assert err.max().item() <= 0.05 * ref.float().abs().max().item() and err.mean().item() <= 4e-3 * ref.float().abs().mean().item() + 1e-3
    # fp32 restatement of the folded product for the v third (no rotation): (x W'^T) * rstd, one bf16 rounding
    v = ((x.float() @ wf[2 * d:].float().t()) * rstd[:, None]).to(torch.bfloat16)
    assert (out[:, 2 * d:].float() - v.float()).abs().max().item() <= 2 ** -7 * v.float().abs().max().item()
    # experts' gate|up: E = 2, rows gathered through slot_token
    E, ff, cap = 2, 1024, 1152
    wg = (torch.randn(E, 2 * ff, d, generator=g) * d ** -0.5).to(torch.bfloat16).to(dev)
    wgf = (wg.float() * ln[None, None, :]).to(torch.bfloat16)
    perm = torch.randperm(T, generator=g)
    slot_token = torch.full((E, cap), 0, dtype=torch.int32)
    slot_token[0, :800] = perm[:800].int(); slot_token[1, :736] = perm[800:].int()
    kept = torch.tensor([800, 736], dtype=torch.int32).to(dev)
    st = slot_token.to(dev)
    a0 = torch.zeros((E, cap, ff), dtype=torch.bfloat16, device=dev); a1 = torch.zeros_like(a0)
    ops.gemm_batched_rows(h, wg, a0, kept, a_rows=st, act=ops.ACT_SWIGLU_PAIR, rows_stride=cap)
    ops.gemm_batched_rows(x, wgf, a1, kept, a_rows=st, act=ops.ACT_SWIGLU_PAIR, rows_stride=cap, a_row_scale=rstd)
    e2 = (a1[0, :800].float() - a0[0, :800].float()).abs()
    print(f"gate|up folded vs unfolded: max {e2.max().item():.3e} mean {e2.mean().item():.3e} ref absmax {a0.float().abs().max().item():.3e}")
    assert e2.max().item() <= 0.06 * a0.float().abs().max().item() and e2.mean().item() <= 6e-3 * a0[0, :800].float().abs().mean().item() + 1e-3
    assert torch.equal(a1[1, 736:], torch.zeros_like(a1[1, 736:]))               # rows beyond an expert's count stay untouched


def test_stack_with_folded_norms_agrees_with_the_unfolded_stack(dev):
    from medplib_amd.model.config import MedPLIBConfig
    from medplib_amd.model.llama import LlamaStack
    kw = dict(num_hidden_layers=2, vocab_size=1024, moe_enable=True)
    a = LlamaStack(MedPLIBConfig.medplib_7b(**kw), dev, seed=5)
    b = LlamaStack(MedPLIBConfig.medplib_7b(fold_input_norm=True, **kw), dev, seed=5)
    g = torch.Generator().manual_seed(9)
    for la, lb in zip(a.layers, b.layers):
        for k in ("ln1", "ln2"):
            v = (1 + 0.3 * torch.randn(la[k].shape, generator=g)).to(dev)
            la[k].copy_(v); lb[k].copy_(v)
    a.refresh_fused_qkv(); b.refresh_fused_qkv()
    assert "qkv_rope_f" in b.layers[0] and "qkv_rope_f" not in a.layers[0]
    x = (torch.randn(3, 512, 4096, generator=g) * 0.5).to(torch.bfloat16).to(dev)
    a.training = b.training = True
    ya, aux_a, ra = a.forward(x, collect_routing=True)
    yb, aux_b, rb = b.forward(x, collect_routing=True)
    assert b.folded_layers == 2 and getattr(a, "folded_layers", 0) == 0
    # (the gate reads the stream BEHIND the attention, whose q / k / v already differ by the fold's rounding: l_aux agrees to that noise, not bit for bit)
    assert abs(float(aux_a[0]) - float(aux_b[0])) < 2e-3 and abs(float(aux_a[1]) - float(aux_b[1])) < 5e-3
    # a random gate has many near-ties: a token that picks the other expert in either layer is a different computation from there on (its row differs
    # by its own magnitude); the comparison is over the rows whose routing agrees in both layers, which must be the large majority
    same = torch.ones(x.shape[0] * x.shape[1], dtype=torch.bool, device=dev)
    for (ea, _, _), (eb, _, _) in zip(ra, rb):
        same &= (ea == eb)
    frac = same.float().mean().item()
    d = ya.shape[-1]
    err = (ya.float() - yb.float()).abs().view(-1, d)[same]
    rel = err.mean().item() / ya.float().abs().view(-1, d)[same].mean().item()
    print(f"2-layer stack folded vs unfolded: routing agrees on {frac:.4f} of the rows; on them mean rel {rel:.3e}, max {err.max().item():.3e} "
          f"(ref absmax {ya.float().abs().max().item():.3e})")
    # two bf16 paths with different rounding points, norm weights spread over 0.1 .. 1.9, and the agreeing rows still attend to the few flipped ones:
    # measured 2.1e-2 (each path alone is ~1e-2 from an fp32 evaluation at this depth, oracle/parity.py's mean bound)
    assert frac > 0.9 and rel < 3e-2
    # B = 1 (639 rows: not a 320-row-kernel shape): the folded model takes the unfolded kernels and is bit-identical with the other stack
    x1 = x[:1, :500].contiguous()
    y1a, _, _ = a.forward(x1); y1b, _, _ = b.forward(x1)
    assert b.folded_layers == 0 and torch.equal(y1a, y1b)
