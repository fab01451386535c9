"""torch.autograd.Function wrappers for the trainable fp32 tail.  torch supplies the tape; every forward and backward
body is HIP kernels from libmedplib_hip.so (no torch arithmetic)."""
import math

import torch

from .. import ops


class LinearFn(torch.autograd.Function):
    """y = act(x @ w^T + b); act in {none, relu}.  nn.Linear / MLP layers (mask_decoder.py:158-186, transformer.py:208-216)."""

    @staticmethod
    def forward(ctx, x, w, b, act):
        x2 = x.reshape(-1, x.shape[-1])
        if not x2.is_contiguous():
            x2 = x2.contiguous()
        y = ops.sgemm(x2, w, trans_b=True, bias=b, act=act)
        ctx.save_for_backward(x2, w, y if act else None)
        ctx.act, ctx.has_bias, ctx.xshape = act, b is not None, x.shape
        return y.view(*x.shape[:-1], w.shape[0])

    @staticmethod
    def backward(ctx, dy):
        x2, w, y = ctx.saved_tensors
        dy2 = dy.reshape(-1, dy.shape[-1])
        if not dy2.is_contiguous():
            dy2 = dy2.contiguous()
        if ctx.act == ops.SACT_RELU:
            dy2 = ops.act_bwd_f32(dy2, y, ops.SACT_RELU)       # y > 0 <=> pre-activation > 0
        dx = ops.sgemm(dy2, w).view(ctx.xshape) if ctx.needs_input_grad[0] else None
        dw = ops.sgemm(dy2, x2, trans_a=True) if ctx.needs_input_grad[1] else None
        db = ops.colsum_f32(dy2) if (ctx.has_bias and ctx.needs_input_grad[2]) else None
        return dx, dw, db, None


class LayerNormFn(torch.autograd.Function):
    """nn.LayerNorm over the last dim (also LayerNorm2d on NHWC tensors, common.py:31-45)."""

    @staticmethod
    def forward(ctx, x, w, b, eps):
        xc = x.contiguous()
        y, mean, rstd = ops.layernorm_fwd_f32(xc, w, b, eps)
        ctx.save_for_backward(xc, w, mean, rstd)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w, mean, rstd = ctx.saved_tensors
        dw = torch.zeros_like(w); db = torch.zeros_like(w)
        dx = ops.layernorm_bwd_f32(dy.contiguous(), x, w, mean, rstd, dw, db)
        return dx, dw, db, None


class AddFn(torch.autograd.Function):
    """a + b (b broadcast over the leading dims when smaller); backward is the identity on both sides (b's gradient is
    only produced when shapes match — broadcast addends here are constants: positional encodings, no_mask_embed)."""

    @staticmethod
    def forward(ctx, a, b):
        ctx.same = a.shape == b.shape
        return ops.add_f32(a.contiguous(), b.contiguous())

    @staticmethod
    def backward(ctx, dy):
        return dy, (dy if ctx.same else None)


class GeluFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        xc = x.contiguous()
        ctx.save_for_backward(xc)
        return ops.act_fwd_f32(xc, ops.SACT_GELU)

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        return ops.act_bwd_f32(dy.contiguous(), x, ops.SACT_GELU)


class AttentionCoreFn(torch.autograd.Function):
    """softmax(q k^T / sqrt(d)) v over heads (transformer.py:224-242); q [B,Nq,C], k/v [B,Nk,C] already projected."""

    @staticmethod
    def forward(ctx, q, k, v, heads):
        B, Nq, C = q.shape
        Nk, d = k.shape[1], C // heads
        q, k, v = q.contiguous(), k.contiguous(), v.contiguous()
        qh = q.view(B, Nq, heads, d).permute(0, 2, 1, 3)
        kh = k.view(B, Nk, heads, d).permute(0, 2, 1, 3)
        vh = v.view(B, Nk, heads, d).permute(0, 2, 1, 3)
        scale = 1.0 / math.sqrt(d)
        s = ops.sgemm(qh, kh, trans_b=True)                       # [B,H,Nq,Nk]
        p = ops.softmax_fwd_f32(s, scale)
        out = torch.empty((B, Nq, C), dtype=torch.float32, device=q.device)
        ops.sgemm(p, vh, out=out.view(B, Nq, heads, d).permute(0, 2, 1, 3))
        ctx.save_for_backward(q, k, v, p)
        ctx.heads, ctx.scale = heads, scale
        return out

    @staticmethod
    def backward(ctx, do):
        q, k, v, p = ctx.saved_tensors
        H = ctx.heads
        B, Nq, C = q.shape
        Nk, d = k.shape[1], C // H
        do = do.contiguous()
        doh = do.view(B, Nq, H, d).permute(0, 2, 1, 3)
        qh = q.view(B, Nq, H, d).permute(0, 2, 1, 3)
        kh = k.view(B, Nk, H, d).permute(0, 2, 1, 3)
        vh = v.view(B, Nk, H, d).permute(0, 2, 1, 3)
        dq = torch.empty_like(q); dk = torch.empty_like(k); dv = torch.empty_like(v)
        ops.sgemm(p, doh, trans_a=True, out=dv.view(B, Nk, H, d).permute(0, 2, 1, 3))     # dV = P^T dO
        dp = ops.sgemm(doh, vh, trans_b=True)                                                 # dP = dO V^T
        ds = ops.softmax_bwd_f32(p, dp, ctx.scale)                                            # includes the 1/sqrt(d)
        ops.sgemm(ds, kh, out=dq.view(B, Nq, H, d).permute(0, 2, 1, 3))                       # dQ = dS K
        ops.sgemm(ds, qh, trans_a=True, out=dk.view(B, Nk, H, d).permute(0, 2, 1, 3))       # dK = dS^T Q
        return dq, dk, dv, None


class ConvT2x2Fn(torch.autograd.Function):
    """ConvTranspose2d(kernel 2, stride 2) on NHWC input as GEMM + pixel shuffle (mask_decoder.py:53-59).
    x [B,h,w,Ci]; weight in the reference layout [Ci,Co,2,2]; returns [B,2h,2w,Co]."""

    @staticmethod
    def forward(ctx, x, w, b):
        B, h, wd, Ci = x.shape
        Co = w.shape[1]
        x2 = x.contiguous().view(B * h * wd, Ci)
        wm = w.view(Ci, Co * 4)
        G = ops.sgemm(x2, wm)
        ctx.save_for_backward(x2, w)
        ctx.dims = (B, h, wd, Ci, Co)
        return ops.convt2x2_shuffle_fwd(G, b, B, h, wd, Co)

    @staticmethod
    def backward(ctx, dy):
        x2, w = ctx.saved_tensors
        B, h, wd, Ci, Co = ctx.dims
        dG = ops.convt2x2_shuffle_bwd(dy.contiguous(), B, h, wd, Co)
        wm = w.view(Ci, Co * 4)
        dx = ops.sgemm(dG, wm, trans_b=True).view(B, h, wd, Ci)
        dw = ops.sgemm(x2, dG, trans_a=True).view(Ci, Co, 2, 2)
        db = ops.colsum_f32(ops.colsum_f32(dG).view(Co, 4).t().contiguous())    # sum over rows, then over the 4 taps
        return dx, dw, db


class HyperDotFn(torch.autograd.Function):
    """masks[b, p] = sum_c hyper[b, c] * up[b, p, c]  (hyper_in @ upscaled_embedding, mask_decoder.py:147-148)."""

    @staticmethod
    def forward(ctx, hyper, up):
        B, P, C = up.shape
        hyper, up = hyper.contiguous(), up.contiguous()
        out = ops.sgemm(up, hyper.view(B, C, 1))          # [B,P,1]
        ctx.save_for_backward(hyper, up)
        return out.view(B, P)

    @staticmethod
    def backward(ctx, dm):
        hyper, up = ctx.saved_tensors
        B, P, C = up.shape
        dm = dm.contiguous().view(B, P, 1)
        dhyper = ops.sgemm(up, dm, trans_a=True).view(B, C)            # up^T dm
        dup = ops.sgemm(dm, hyper.view(B, 1, C))                       # outer product
        return dhyper, dup


class FusedUpsampleMaskFn(torch.autograd.Function):
    """output_upscaling + hyper_in @ upscaled_embedding (mask_decoder.py:53-59,141-148) as the single fused bf16 kernel, differentiable:
    ConvT(256->64) -> LayerNorm2d -> GELU -> ConvT(64->32) -> GELU -> . hyper, in the arithmetic the reference itself trains in under
    `--precision bf16` (bf16 operands, fp32 accumulation).  src [n, g*g, 256] fp32 tokens (NHWC order), weights in the reference layout
    [Cin, Cout, 2, 2], hyper [n, 32]  ->  mask logits [n, 4g, 4g] fp32.  The backward is one recomputing kernel
    (mp_mask_upsample_fused_bwd_bf16) + the two weight-gradient `tn` GEMMs + fixed-order column sums: deterministic."""

    @staticmethod
    def forward(ctx, src, w1, b1, lnw, lnb, w2, b2, hyper, g, eps):
        n = src.shape[0]
        src = src.contiguous()
        src_bf = ops.cast_to_bf16(src).view(n, g * g, 256)
        w1p, w2p = ops.pack_upsampler_weights(w1.detach(), w2.detach())
        b1, lnw, lnb, b2 = (t.detach().float().contiguous() for t in (b1, lnw, lnb, b2))
        hyper = hyper.detach().contiguous()
        _, masks = ops.mask_upsample_fused(src_bf, w1p, b1, lnw, lnb, w2p, b2, g, g, hyper=hyper, want_up=False, eps=eps)
        ctx.save_for_backward(src_bf, w1p, w2p, b1, lnw, lnb, b2, hyper)
        ctx.dims = (n, g, eps, tuple(src.shape))
        return masks

    @staticmethod
    def backward(ctx, dm):
        src_bf, w1p, w2p, b1, lnw, lnb, b2, hyper = ctx.saved_tensors
        n, g, eps, src_shape = ctx.dims
        need = ctx.needs_input_grad
        dx2, dy1, a1, dy2, part = ops.mask_upsample_fused_bwd(src_bf, w1p, b1, lnw, lnb, w2p, b2, hyper, dm.contiguous().float(), g, g, eps=eps)
        dx = ops.add_f32(dx2[0], dx2[1]).view(src_shape) if need[0] else None
        # The weight gradient of the function that was EVALUATED: the forward (and the recomputing backward) consumed the bf16-rounded
        # tokens, so dW1 = dy1^T . src_bf, not . src (round-3 advisor); the fp32 `src` is no longer kept for the backward at all.
        # Packed rows are (kh, kw, cout): back to the reference layout [Cin, Cout, 2, 2].  A frozen mask decoder (needs_input_grad false
        # for every parameter: stage-II-style runs, inference-time gradients w.r.t. the tokens only) skips the GEMMs and column sums.
        dw1 = dw2 = None
        if need[1]:
            dw1 = ops.sgemm(dy1, ops.cast_to_f32(src_bf).view(-1, 256), trans_a=True).view(2, 2, 64, 256).permute(3, 2, 0, 1).contiguous()
        if need[5]:
            dw2 = ops.sgemm(dy2, a1, trans_a=True).view(2, 2, 32, 64).permute(3, 2, 0, 1).contiguous()
        cs = ops.colsum_f32(part) if any(need[i] for i in (2, 3, 4, 6)) else None
        pick = (lambda i, a, b: cs[a:b] if need[i] else None)
        dhyper = part[:, 224:].reshape(n, -1, 32).sum(1) if need[7] else None
        return dx, dw1, pick(2, 0, 64), pick(3, 64, 128), pick(4, 128, 192), dw2, pick(6, 192, 224), dhyper, None, None


class BilinearResizeFn(torch.autograd.Function):
    """postprocess_masks (MedPLIB.py:682-701) for n masks sharing one (input_size, original_size)."""

    @staticmethod
    def forward(ctx, low_res, crop, out_hw):
        ctx.crop, ctx.in_hw = crop, tuple(low_res.shape[-2:])
        return ops.bilinear_resize_fwd(low_res.contiguous(), crop, out_hw)

    @staticmethod
    def backward(ctx, dy):
        return ops.bilinear_resize_bwd(dy.contiguous(), ctx.in_hw, ctx.crop), None, None


class MaskLossFn(torch.autograd.Function):
    """All four mask losses + the weighted combination in one pass (MedPLIB.py:515-572).  Returns the 10 scalars of the
    reference's output dict as one tensor; only out[0] ('loss') is differentiable."""

    @staticmethod
    def forward(ctx, pred, gt, pred_iou, ce_loss, weights, offsets=None):
        pred, gt, pred_iou = pred.contiguous(), gt.contiguous(), pred_iou.contiguous()
        out, stats = ops.mask_losses_fwd(pred, gt, pred_iou, ce_loss, weights, offsets)
        ctx.save_for_backward(pred, gt, stats)
        ctx.weights, ctx.offsets = weights, offsets
        return out

    @staticmethod
    def backward(ctx, dout):
        pred, gt, stats = ctx.saved_tensors
        gscale = dout[0:1].contiguous()
        dpred, dq = ops.mask_losses_bwd(pred, gt, stats, gscale, ctx.weights, ctx.offsets)
        d_ce = gscale * ctx.weights[0] if ctx.needs_input_grad[3] else None      # loss = ce * ce_loss_weight + mask terms
        return dpred, None, dq, d_ce, None, None


class BuildTokensFn(torch.autograd.Function):
    """tokens = cat([iou_token; mask_tokens] expanded over the batch, text_embeds) (mask_decoder.py:123-125)."""

    @staticmethod
    def forward(ctx, iou_w, mask_w, text):
        B, C = text.shape[0], text.shape[-1]
        tok = torch.empty((B, 6, C), dtype=torch.float32, device=text.device)
        tok[:, 0:1].copy_(iou_w.unsqueeze(0).expand(B, -1, -1))
        tok[:, 1:5].copy_(mask_w.unsqueeze(0).expand(B, -1, -1))
        tok[:, 5:6].copy_(text)
        return tok

    @staticmethod
    def backward(ctx, dtok):
        B, _, C = dtok.shape
        d5 = ops.colsum_f32(dtok[:, :5].contiguous().view(B, 5 * C)).view(5, C)
        return d5[0:1], d5[1:5], dtok[:, 5:6]


def linear(x, w, b=None, act=ops.SACT_NONE):
    return LinearFn.apply(x, w, b, act)


def layernorm(x, w, b, eps=1e-5):
    return LayerNormFn.apply(x, w, b, eps)


def add(a, b):
    return AddFn.apply(a, b)
