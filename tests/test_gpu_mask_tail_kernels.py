"""GPU parity: fp32 tail kernels (sgemm, LayerNorm/softmax fwd+bwd, ConvT shuffle, gathers) and the mask-head kernels
(postprocess resize, fused losses fwd+bwd, threshold/IoU) through the C ABI vs the CPU oracle and the golden vectors
generated from the reference (tests/golden/mask_head_reference.npz).

Tolerances: fp32 kernels vs fp32 oracle — 1e-5 relative (accumulation order / FMA contraction only); integer outputs
(threshold mask, counts) bit-exact."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import ops as O

pytestmark = pytest.mark.gpu


def _close(name, got, ref, rtol=1e-5, atol=1e-5):
    got, ref = got.float().cpu(), ref.float().cpu()
    err = (got - ref).abs()
    bad = err > atol + rtol * ref.abs()
    msg = f"{name}: max|err|={err.max().item():.3e}, ref absmax={ref.abs().max().item():.3e}, bad={int(bad.sum())}/{bad.numel()}"
    print(msg)
    assert not bad.any(), msg


def test_sgemm_forms(dev):
    from medplib_amd import ops
    g = torch.Generator().manual_seed(1)
    for (M, N, K) in [(6, 256, 256), (2048, 128, 256), (8, 4096, 4096), (70, 33, 19)]:
        a = torch.randn(M, K, generator=g); b = torch.randn(K, N, generator=g); bias = torch.randn(N, generator=g)
        _close(f"NN {M},{N},{K}", ops.sgemm(a.to(dev), b.to(dev)), a @ b, rtol=1e-4, atol=1e-4 * K ** 0.5)
        _close("NT+bias+relu", ops.sgemm(a.to(dev), b.T.contiguous().to(dev), trans_b=True, bias=bias.to(dev), act=ops.SACT_RELU),
               torch.relu(a @ b + bias), rtol=1e-4, atol=1e-4 * K ** 0.5)
        _close("TN", ops.sgemm(a.T.contiguous().to(dev), b.to(dev), trans_a=True), a @ b, rtol=1e-4, atol=1e-4 * K ** 0.5)
    # split-K accumulate into a pre-initialised C
    a = torch.randn(8, 4096, generator=g); b = torch.randn(4096, 512, generator=g); c0 = torch.randn(8, 512, generator=g)
    c = c0.clone().to(dev)
    ops.sgemm(a.to(dev), b.to(dev), out=c, beta=1.0, split_k=8)
    _close("split_k", c, c0 + a @ b, rtol=1e-4, atol=1e-2)
    # two-level batch over (batch, heads) with strided head views: scores = q k^T
    B, Nq, Nk, H, d = 3, 6, 256, 8, 16
    q = torch.randn(B, Nq, H * d, generator=g); k = torch.randn(B, Nk, H * d, generator=g)
    qd, kd = q.to(dev), k.to(dev)
    qv = qd.view(B, Nq, H, d).permute(0, 2, 1, 3)   # [B,H,Nq,d] strided
    kv = kd.view(B, Nk, H, d).permute(0, 2, 1, 3)
    s = ops.sgemm(qv, kv, trans_b=True, alpha=0.25)
    ref = torch.einsum("bqhd,bkhd->bhqk", q.view(B, Nq, H, d), k.view(B, Nk, H, d)) * 0.25
    _close("batched heads NT", s, ref, rtol=1e-4, atol=1e-4)


_SGEMM_AB = r"""
import os, sys, torch
sys.path.insert(0, os.environ["REPO"])
from medplib_amd import ops
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(9)
out = {}
for i, (M, N, K, nb) in enumerate([(2048, 256, 256, 0), (70, 33, 19, 0), (2048, 2048, 256, 0), (8, 256, 4096, 0), (300, 130, 66, 3), (64, 64, 16, 0)]):
    bs = (nb,) if nb else ()
    a = torch.randn(*bs, M, K, generator=g).to(dev); b = torch.randn(*bs, K, N, generator=g).to(dev); bias = torch.randn(N, generator=g).to(dev)
    out[f"nn{i}"] = ops.sgemm(a, b).cpu()
    out[f"nt{i}"] = ops.sgemm(a, b.transpose(-1, -2).contiguous(), trans_b=True, bias=bias, act=ops.SACT_GELU).cpu()
    out[f"tn{i}"] = ops.sgemm(a.transpose(-1, -2).contiguous(), b, trans_a=True, alpha=0.37).cpu()
torch.save(out, os.environ["OUT"])
"""


def test_sgemm_mfma_form_is_bitwise_the_fma_chain(dev, tmp_path):
    """The tail's fp32 GEMM runs its inner product on v_mfma_f32_32x32x2_f32 (round 3): same staging, same ascending-k accumulation —
    the hardware's f32 MFMA is an fmaf chain — so every output must equal the register-tile FMA loop (MP_SGEMM_MFMA=0) BIT FOR BIT:
    NN / NT (+ bias + GELU) / TN (alpha), ragged edges, a long-K skinny product (the deterministic split), a batched call."""
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "sgemm_ab.py"
    script.write_text(_SGEMM_AB)
    res = {}
    for mode in ("1", "0"):
        outp = tmp_path / f"out{mode}.pt"
        env = dict(os.environ, REPO=repo, OUT=str(outp), MP_SGEMM_MFMA=mode)
        p = subprocess.run([sys.executable, str(script)], env=env, capture_output=True, text=True, timeout=600)
        assert p.returncode == 0, p.stderr[-2000:]
        res[mode] = torch.load(outp)
    assert set(res["1"]) == set(res["0"]) and len(res["1"]) == 18
    for k in res["1"]:
        assert torch.equal(res["1"][k], res["0"][k]), (k, (res["1"][k] - res["0"][k]).abs().max().item())


def test_layernorm_softmax_act_fwd_bwd(dev):
    from medplib_amd import ops
    g = torch.Generator().manual_seed(2)
    for rows, dim in [(48, 256), (2048, 64), (5, 4096)]:
        x = torch.randn(rows, dim, generator=g, requires_grad=True)
        w = (1 + 0.1 * torch.randn(dim, generator=g)).requires_grad_(); b = (0.1 * torch.randn(dim, generator=g)).requires_grad_()
        dy = torch.randn(rows, dim, generator=g)
        y = F.layer_norm(x, (dim,), w, b, 1e-5); y.backward(dy)
        yd, mean, rstd = ops.layernorm_fwd_f32(x.detach().to(dev), w.detach().to(dev), b.detach().to(dev), 1e-5)
        _close(f"ln fwd {rows}x{dim}", yd, y.detach())
        dw = torch.zeros(dim, device=dev); db = torch.zeros(dim, device=dev)
        dx = ops.layernorm_bwd_f32(dy.to(dev), x.detach().to(dev), w.detach().to(dev), mean, rstd, dw, db)
        _close("ln dx", dx, x.grad, rtol=1e-4, atol=1e-5)
        _close("ln dw", dw, w.grad, rtol=1e-4, atol=1e-4)
        _close("ln db", db, b.grad, rtol=1e-4, atol=1e-4)
    x = torch.randn(96, 256, generator=g, requires_grad=True); dp = torch.randn(96, 256, generator=g)
    p = torch.softmax(x * 0.25, -1); p.backward(dp)
    pd = ops.softmax_fwd_f32(x.detach().to(dev), 0.25)
    _close("softmax fwd", pd, p.detach(), rtol=1e-5, atol=1e-7)
    _close("softmax bwd", ops.softmax_bwd_f32(pd, dp.to(dev), 0.25), x.grad, rtol=1e-4, atol=1e-7)
    for act, fn in ((ops.SACT_RELU, torch.relu), (ops.SACT_GELU, F.gelu)):
        x = torch.randn(1000, generator=g, requires_grad=True); dy = torch.randn(1000, generator=g)
        y = fn(x); y.backward(dy)
        _close(f"act{act} fwd", ops.act_fwd_f32(x.detach().to(dev), act), y.detach())
        _close(f"act{act} bwd", ops.act_bwd_f32(dy.to(dev), x.detach().to(dev), act), x.grad, rtol=1e-4, atol=1e-6)
    x = torch.randn(777, 130, generator=g)
    _close("colsum", ops.colsum_f32(x.to(dev)), x.sum(0), rtol=1e-4, atol=1e-4)
    a = torch.randn(4, 6, 8, generator=g); bb = torch.randn(6, 8, generator=g)
    _close("add_f32 bcast", ops.add_f32(a.to(dev), bb.to(dev)), a + bb)


def test_convt2x2_as_gemm_plus_shuffle(dev):
    """ConvTranspose2d(k=2,s=2) == [B*h*w, Cin] @ W.view(Cin, Cout*4) + pixel shuffle (mask_decoder.py:53-59)."""
    from medplib_amd import ops
    g = torch.Generator().manual_seed(3)
    B, Ci, Co, h, w = 2, 256, 64, 16, 16
    x = torch.randn(B, Ci, h, w, generator=g, requires_grad=True)
    W = (torch.randn(Ci, Co, 2, 2, generator=g) * 0.05).requires_grad_(); bias = torch.randn(Co, generator=g)
    y = F.conv_transpose2d(x, W, bias, stride=2)
    dy = torch.randn_like(y); y.backward(dy)
    xt = x.detach().permute(0, 2, 3, 1).reshape(B * h * w, Ci).contiguous().to(dev)
    Wm = W.detach().reshape(Ci, Co * 4).to(dev)
    G = ops.sgemm(xt, Wm)
    Y = ops.convt2x2_shuffle_fwd(G, bias.to(dev), B, h, w, Co)
    _close("convT fwd", Y, y.detach().permute(0, 2, 3, 1), rtol=1e-4, atol=1e-4)
    dG = ops.convt2x2_shuffle_bwd(dy.permute(0, 2, 3, 1).contiguous().to(dev), B, h, w, Co)
    dX = ops.sgemm(dG, Wm, trans_b=True)
    _close("convT dX", dX, x.grad.permute(0, 2, 3, 1).reshape(B * h * w, Ci), rtol=1e-4, atol=1e-4)
    dW = ops.sgemm(xt, dG, trans_a=True)
    _close("convT dW", dW, W.grad.reshape(Ci, Co * 4), rtol=1e-4, atol=1e-3)


def test_gathers(dev):
    from medplib_amd import ops
    g = torch.Generator().manual_seed(4)
    src = torch.randn(5, 9, 64, generator=g).to(torch.bfloat16)
    idx = torch.tensor([44, 0, 13, 13], dtype=torch.int64)
    out = ops.gather_rows_bf16_to_f32(src.to(dev), idx.to(dev))
    assert torch.equal(out.cpu(), src.view(-1, 64)[idx].float())
    emb = torch.randn(3, 4, 5, generator=g)
    out = ops.gather_rows_f32(emb.to(dev), torch.tensor([2, 2, 0], device=dev))
    assert torch.equal(out.cpu(), emb[[2, 2, 0]])


def test_postprocess_resize_against_reference_golden(dev, golden_dir):
    """postprocess_masks incl. the reference's Python-slice crop quirks; goldens come from model/MedPLIB.py itself."""
    from medplib_amd import ops
    gz = np.load(os.path.join(golden_dir, "mask_head_reference.npz"))
    for i in range(int(gz["pp_count"])):
        x = torch.from_numpy(gz[f"pp{i}_in"])[0]                 # [1,64,64]
        inp = tuple(int(v) for v in gz[f"pp{i}_input_size"]); orig = tuple(int(v) for v in gz[f"pp{i}_original_size"])
        crop = ops.postprocess_crop(64, 64, inp)
        out = ops.bilinear_resize_fwd(x.contiguous().to(dev), crop, orig)
        # fp32 interpolation with identical taps; the source coordinate scale*(dst+0.5)-0.5 is evaluated with / without
        # FMA contraction on the two sides, so the lerp weight differs by ~ulp(63) = 4e-6 -> 3e-5 absolute on O(1) logits
        _close(f"postprocess case {i} {inp}->{orig} crop={crop}", out, torch.from_numpy(gz[f"pp{i}_out"])[0], rtol=1e-6, atol=3e-5)
    # backward vs autograd
    x = torch.randn(3, 64, 64, requires_grad=True)
    y = F.interpolate(x[:, None, :, 12:52], (77, 50), mode="bilinear", align_corners=False)
    dy = torch.randn_like(y); y.backward(dy)
    din = ops.bilinear_resize_bwd(dy[:, 0].contiguous().to(dev), (64, 64), (0, 12, 64, 40))
    _close("bilinear bwd", din, x.grad, rtol=1e-4, atol=1e-5)


def test_mask_losses_fwd_bwd(dev, golden_dir):
    from medplib_amd import ops
    gz = np.load(os.path.join(golden_dir, "mask_head_reference.npz"))
    pred = torch.from_numpy(gz["loss_pred"]); gt = torch.from_numpy(gz["loss_gt"]); piou = torch.from_numpy(gz["loss_pred_iou"])
    n, _, H, W = pred.shape
    weights = dict(ce=1.0, bce=2.0, dice=0.5, iou=1.5, focal=3.0)
    wl = (weights["ce"], weights["bce"], weights["dice"], weights["iou"], weights["focal"])
    ce = torch.tensor([0.37])
    p = pred.clone().requires_grad_(); q = piou.clone().requires_grad_()
    ref = O.combine_mask_losses([p[i] for i in range(n)], [gt[i] for i in range(n)], [q[i] for i in range(n)], ce[0], weights)
    ref["loss"].backward()
    out, stats = ops.mask_losses_fwd(pred.view(n, -1).contiguous().to(dev), gt.view(n, -1).contiguous().to(dev),
                                     piou.view(-1).contiguous().to(dev), ce.to(dev), wl)
    for i, k in enumerate(O.LOSS_KEYS):
        _close(f"loss[{k}]", out[i], ref[k].detach(), rtol=2e-5, atol=1e-6)
    # per-mask terms pinned by the reference golden: sum over masks / (n + 1e-8)
    terms = gz["loss_terms"]
    _close("unscaled bce vs reference golden", out[5], torch.tensor(terms[:, 0].sum() / (n + 1e-8)), rtol=2e-5, atol=1e-6)
    _close("unscaled focal vs reference golden", out[9], torch.tensor(terms[:, 3].sum() / (n + 1e-8)), rtol=2e-5, atol=1e-6)
    dpred, dq = ops.mask_losses_bwd(pred.view(n, -1).contiguous().to(dev), gt.view(n, -1).contiguous().to(dev), stats, None, wl)
    _close("dloss/dpred", dpred.view_as(pred), p.grad, rtol=1e-3, atol=1e-8)
    _close("dloss/dpred_iou", dq, q.grad.view(-1), rtol=1e-4, atol=1e-7)


def test_mask_losses_ragged_sizes(dev):
    """Masks of different H x W in one launch (flat buffers + offsets) vs the oracle's per-mask loop (MedPLIB.py:515-559)."""
    from medplib_amd import ops
    g = torch.Generator().manual_seed(17)
    shapes = [(96, 80), (40, 131), (336, 336), (7, 5)]
    preds = [(torch.randn(1, h, w, generator=g) * 3).requires_grad_() for h, w in shapes]
    gts = [(torch.rand(h, w, generator=g) > 0.6).float() for h, w in shapes]
    qs = [torch.rand(1, generator=g).requires_grad_() for _ in shapes]
    weights = dict(ce=1.0, bce=2.0, dice=0.5, iou=1.5, focal=3.0)
    wl = (1.0, 2.0, 0.5, 1.5, 3.0)
    ce = torch.tensor([0.21])
    ref = O.combine_mask_losses(preds, gts, qs, ce[0], weights)
    ref["loss"].backward()
    sizes = [h * w for h, w in shapes]
    off = torch.tensor([0] + list(np.cumsum(sizes)), dtype=torch.int64, device=dev)
    pf = torch.cat([p.detach().reshape(-1) for p in preds]).to(dev); gf = torch.cat([x.reshape(-1) for x in gts]).to(dev)
    qd = torch.cat([q.detach() for q in qs]).to(dev)
    out, stats = ops.mask_losses_fwd(pf, gf, qd, ce.to(dev), wl, offsets=off)
    for i, k in enumerate(O.LOSS_KEYS):
        _close(f"ragged loss[{k}]", out[i], ref[k].detach(), rtol=2e-5, atol=1e-6)
    dpred, dq = ops.mask_losses_bwd(pf, gf, stats, None, wl, offsets=off)
    _close("ragged dloss/dpred", dpred, torch.cat([p.grad.reshape(-1) for p in preds]), rtol=1e-3, atol=1e-8)
    _close("ragged dloss/dpred_iou", dq, torch.cat([q.grad for q in qs]), rtol=1e-4, atol=1e-7)


def test_threshold_iou_bit_exact(dev, golden_dir):
    from medplib_amd import ops
    gz = np.load(os.path.join(golden_dir, "mask_head_reference.npz"))
    pred = torch.from_numpy(gz["loss_pred"])[:, 0]; gt = torch.from_numpy(gz["loss_gt"])
    n = pred.shape[0]
    b, counts = ops.mask_threshold_iou(pred.reshape(n, -1).contiguous().to(dev), gt.reshape(n, -1).contiguous().to(dev), 0.1)
    assert np.array_equal(b[0].cpu().numpy().astype(bool).reshape(gz["thr_mask"].shape), gz["thr_mask"]), "mask indices must be bit-exact"
    assert counts[0].cpu().tolist() == gz["thr_counts"].tolist()
    for i in range(n):
        rb, rc, _, _ = O.threshold_iou(pred[i], gt[i])
        assert torch.equal(b[i].cpu().bool().view_as(rb), rb) and counts[i].cpu().tolist() == list(rc)
    # full BASELINE size, size-independent property: counts consistent (|and| + |or| = |pred| + |gt|)
    big = torch.randn(8, 336 * 336) * 4
    gtb = (torch.rand(8, 336 * 336) > 0.5).float()
    b, c = ops.mask_threshold_iou(big.to(dev), gtb.to(dev), 0.1)
    c = c.cpu()
    assert torch.equal(c[:, 2] + c[:, 3], c[:, 0] + c[:, 1])
    assert torch.equal(b.cpu().bool(), torch.sigmoid(big) > 0.1)


@pytest.mark.parametrize("grid,B", [(16, 8), (64, 2)])
def test_fused_upsampler_matches_reference_chain(dev, grid, B):
    """mask_decoder.output_upscaling + hyper product as ONE kernel (bf16), at the model's 256-px geometry (16x16 tokens) and at
    the 1024-px SAM geometry (64x64 tokens), against the oracle's conv_transpose2d / LayerNorm2d / GELU chain run in fp32 on
    the same bf16-rounded inputs.  Tolerance: intermediate activations are rounded to bf16 once (2 ulps of O(1) values)."""
    from medplib_amd import ops
    from oracle import sam as OS
    W = OS.init_weights(seed=5)
    g = torch.Generator().manual_seed(grid)
    bf = lambda t: t.to(torch.bfloat16).float()
    src = bf(torch.randn(B, 256, grid, grid, generator=g))
    Wq = dict(W)
    Wq["mask_decoder.output_upscaling.0.weight"] = bf(W["mask_decoder.output_upscaling.0.weight"])
    Wq["mask_decoder.output_upscaling.3.weight"] = bf(W["mask_decoder.output_upscaling.3.weight"])
    ref_up = OS.output_upscaling(src, Wq)                                        # [B,32,4g,4g]
    hyper = torch.randn(B, 32, generator=g)
    ref_mask = (hyper[:, None, :] @ bf(ref_up).view(B, 32, -1)).view(B, 4 * grid, 4 * grid)
    w1p, w2p = ops.pack_upsampler_weights(W["mask_decoder.output_upscaling.0.weight"].to(dev), W["mask_decoder.output_upscaling.3.weight"].to(dev))
    tok = src.permute(0, 2, 3, 1).reshape(B, grid * grid, 256).contiguous().to(torch.bfloat16).to(dev)
    d = lambda k: W[k].to(dev)
    up, mask = ops.mask_upsample_fused(tok, w1p, d("mask_decoder.output_upscaling.0.bias"), d("mask_decoder.output_upscaling.1.weight"),
                                       d("mask_decoder.output_upscaling.1.bias"), w2p, d("mask_decoder.output_upscaling.3.bias"),
                                       grid, grid, hyper=hyper.to(dev))
    _close(f"fused upsampler up (grid {grid})", up, ref_up, rtol=2 ** -7, atol=2e-2)
    _close(f"fused upsampler mask (grid {grid})", mask, ref_mask, rtol=1e-2, atol=8e-2)


@pytest.mark.parametrize("h,w,B", [(64, 64, 8), (64, 64, 9), (8, 48, 5), (128, 64, 3), (1, 16, 1)])
def test_fused_upsampler_large_launches_and_output_forms(dev, h, w, B):
    """The launch shapes the two cases above do not reach: sixteen waves per workgroup with one pass per wave (64 x 64 x 8 = 2048 groups: the
    benchmark's geometry, the first four waves' tokens requested before the staging wait), more groups than 256 workgroups x 8 (a second pass
    for some waves: 2304 and 1536 x ... groups), non-square maps, a group count that leaves waves without work, one group.  Against the fp32
    reference chain, and the three output forms against each other: the `up` of (up, mask) and of up alone, the mask of (up, mask) and of
    mask alone are the same bits (same arithmetic, only the stores differ)."""
    from medplib_amd import ops
    from oracle import sam as OS
    W = OS.init_weights(seed=7)
    g = torch.Generator().manual_seed(h * 1000 + w + B)
    bf = lambda t: t.to(torch.bfloat16).float()
    src = bf(torch.randn(B, 256, h, w, generator=g))
    Wq = dict(W)
    Wq["mask_decoder.output_upscaling.0.weight"] = bf(W["mask_decoder.output_upscaling.0.weight"])
    Wq["mask_decoder.output_upscaling.3.weight"] = bf(W["mask_decoder.output_upscaling.3.weight"])
    hyper = torch.randn(B, 32, generator=g)
    w1p, w2p = ops.pack_upsampler_weights(W["mask_decoder.output_upscaling.0.weight"].to(dev), W["mask_decoder.output_upscaling.3.weight"].to(dev))
    tok = src.permute(0, 2, 3, 1).reshape(B, h * w, 256).contiguous().to(torch.bfloat16).to(dev)
    d = lambda k: W[k].to(dev)
    args = (tok, w1p, d("mask_decoder.output_upscaling.0.bias"), d("mask_decoder.output_upscaling.1.weight"),
            d("mask_decoder.output_upscaling.1.bias"), w2p, d("mask_decoder.output_upscaling.3.bias"), h, w)
    up, mask = ops.mask_upsample_fused(*args, hyper=hyper.to(dev))
    up_only, none = ops.mask_upsample_fused(*args)
    none2, mask_only = ops.mask_upsample_fused(*args, hyper=hyper.to(dev), want_up=False)
    torch.cuda.synchronize()
    assert none is None and none2 is None
    assert torch.equal(up, up_only), "up of (up, mask) vs up alone"
    assert torch.equal(mask, mask_only), "mask of (up, mask) vs mask alone"
    n_ref = min(B, 2)                                             # the fp32 chain on the host for the first and the last image
    for b in sorted({0, B - 1})[:n_ref]:
        ref_up = OS.output_upscaling(src[b:b + 1], Wq)
        ref_mask = (hyper[b:b + 1, None, :] @ bf(ref_up).view(1, 32, -1)).view(1, 4 * h, 4 * w)
        _close(f"fused upsampler up ({h}x{w}x{B}, image {b})", up[b:b + 1], ref_up, rtol=2 ** -7, atol=2e-2)
        _close(f"fused upsampler mask ({h}x{w}x{B}, image {b})", mask[b:b + 1], ref_mask, rtol=1e-2, atol=8e-2)
    again, _ = ops.mask_upsample_fused(*args)
    assert torch.equal(again, up_only), "same bits on every launch"


def test_fused_upsampler_backward_vs_fp32_autograd(dev):
    """The training form of the fused upsampler (A.FusedUpsampleMaskFn: forward = mp_mask_upsample_fused_bf16 with the hypernetwork
    product, backward = mp_mask_upsample_fused_bwd_bf16 + two `tn` GEMMs + column sums) against torch's own fp32 ConvTranspose2d ->
    LayerNorm2d -> GELU -> ConvTranspose2d -> GELU -> hyper product and its autograd, on the model's geometry (16 x 16 tokens) with 3
    prompts.  bf16 operands with fp32 accumulation: logits within 2e-2 of their scale, every gradient within 3e-2 of its own scale in
    norm (measured ~5e-3); two runs are bit-identical (no atomics)."""
    import torch.nn.functional as F
    from medplib_amd.model import autograd_ops as A
    _fused_upsampler_training_case(dev, n=3, G=16, seed=41)


def test_fused_upsampler_backward_many_groups_per_wave(dev):
    """The same check at the 1024-px geometry (64 x 64 tokens, 3 prompts = 768 sixteen-token groups): more groups than the launch has
    waves, so every wave of the forward and of the backward walks several groups (the persistent loops' stride) and a workgroup's partial
    rows cover more than one image."""
    _fused_upsampler_training_case(dev, n=3, G=64, seed=43)


def _fused_upsampler_training_case(dev, n, G, seed):
    import torch.nn.functional as F
    from medplib_amd.model import autograd_ops as A
    g = torch.Generator().manual_seed(seed)
    src = torch.randn(n, G * G, 256, generator=g)
    w1 = torch.randn(256, 64, 2, 2, generator=g) * 0.06; b1 = torch.randn(64, generator=g) * 0.1
    lnw = 1.0 + 0.2 * torch.randn(64, generator=g); lnb = 0.1 * torch.randn(64, generator=g)
    w2 = torch.randn(64, 32, 2, 2, generator=g) * 0.12; b2 = torch.randn(32, generator=g) * 0.1
    hyper = torch.randn(n, 32, generator=g) * 0.5
    dm = torch.randn(n, 4 * G, 4 * G, generator=g)
    names = ["src", "w1", "b1", "lnw", "lnb", "w2", "b2", "hyper"]

    def reference(ts):
        s_, w1_, b1_, lw_, lb_, w2_, b2_, h_ = ts
        x = s_.view(n, G, G, 256).permute(0, 3, 1, 2)
        y = F.conv_transpose2d(x, w1_, b1_, stride=2)
        u = y.mean(1, keepdim=True); v = (y - u).pow(2).mean(1, keepdim=True)
        y = (y - u) / torch.sqrt(v + 1e-6) * lw_[None, :, None, None] + lb_[None, :, None, None]
        y = F.gelu(F.conv_transpose2d(F.gelu(y), w2_, b2_, stride=2))
        return torch.einsum("bc,bchw->bhw", h_, y)

    ref_in = [t.clone().requires_grad_() for t in (src, w1, b1, lnw, lnb, w2, b2, hyper)]
    ref_out = reference(ref_in)
    ref_out.backward(dm)
    runs = []
    for _ in range(2):
        ins = [t.to(dev).clone().requires_grad_() for t in (src, w1, b1, lnw, lnb, w2, b2, hyper)]
        out = A.FusedUpsampleMaskFn.apply(*ins, G, 1e-6)
        out.backward(dm.to(dev))
        torch.cuda.synchronize()
        runs.append((out.detach(), [t.grad.detach() for t in ins]))
    assert torch.equal(runs[0][0], runs[1][0]) and all(torch.equal(a, b) for a, b in zip(runs[0][1], runs[1][1]))
    out, grads = runs[0]
    scale = float(ref_out.abs().max())
    err = float((out.cpu() - ref_out.detach()).abs().max())
    print(f"fused upsampler (training form) logits: max err {err:.3e} of scale {scale:.3e}")
    assert err < 2e-2 * scale
    for name, got, ref in zip(names, grads, ref_in):
        assert got.shape == ref.grad.shape, name
        rel = float((got.cpu() - ref.grad).norm() / ref.grad.norm())
        print(f"fused upsampler d{name}: relative Frobenius error {rel:.3e}")
        assert rel < 3e-2, (name, rel)
