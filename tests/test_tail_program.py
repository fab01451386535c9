"""The mask-tail program (medplib_amd/tail_program.py -> csrc/tail_program.hip), CPU half: the LOWERING — op table, phases, the generated
backward — is executed by the numpy interpreter of the same packed table (oracle/tail_program_emu.py) on host memory and compared with
torch autograd over the oracle's mask decoder (oracle/sam.py, which follows transformer.py:62-106,151-244 and mask_decoder.py:113-153) and
text_hidden_fcs (MedPLIB.py:152-164).  The HIP kernel's half is tests/test_gpu_tail_program.py."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from medplib_amd import tail_program as TP
from medplib_amd.model.sam import MaskDecoder, PromptEncoderText
from oracle import sam as OS
from oracle import tail_program_emu as EMU


def _setup(n, Dh, seed, with_fcs=True):
    torch.manual_seed(seed)
    W = OS.init_weights(seed=seed)
    dec, pe = MaskDecoder(), PromptEncoderText()
    dec.load_state_dict({k[len("mask_decoder."):]: v for k, v in W.items() if k.startswith("mask_decoder.")})
    pe.load_state_dict({k[len("prompt_encoder."):]: v for k, v in W.items() if k.startswith("prompt_encoder.")}, strict=False)
    fcs = (torch.nn.Linear(Dh, Dh), torch.nn.Linear(Dh, 256)) if with_fcs else None
    params = (list(fcs[0].parameters()) + list(fcs[1].parameters()) if with_fcs else []) + list(dec.parameters())
    offs, tot = {}, 0
    for p in params:
        offs[id(p)] = tot * 4
        tot += p.numel()
    return W, dec, pe, fcs, params, offs, tot


def _reference(W, dec, fcs, x_in, img, n, d_src, d_hy, d_iou):
    Wr = {k: v.clone() for k, v in W.items()}
    for k, v in dec.named_parameters():
        Wr["mask_decoder." + k] = v
    xr = x_in.clone().requires_grad_()
    text = (fcs[1](F.relu(fcs[0](xr))) if fcs is not None else xr).view(n, 1, 256)
    sp, de = OS.prompt_encoder_text(text, Wr)
    emb = img.view(n, 16, 16, 256).permute(0, 3, 1, 2)
    _, _, _, hs, src2 = OS.mask_decoder(emb, OS.dense_pe(Wr), sp, de, Wr, return_all=True)
    hy = OS._mlp3(hs[:, 1], Wr, "mask_decoder.output_hypernetworks_mlps.0")
    iou = OS._mlp3(hs[:, 0], Wr, "mask_decoder.iou_prediction_head")
    ((src2 * (d_src[0] + d_src[1])).sum() + (hy * d_hy).sum() + (iou[:, 0] * d_iou).sum()).backward()
    return src2.detach(), hy.detach(), iou.detach(), xr.grad


@pytest.mark.parametrize("n,with_fcs", [(1, True), (3, True), (2, False)])
def test_lowering_matches_oracle_autograd(n, with_fcs):
    Dh = 320
    W, dec, pe, fcs, params, offs, tot = _setup(n, Dh, seed=3 + n, with_fcs=with_fcs)
    prog = TP.TailProgram(dec, n, pe.dense_pe_tokens(), pe.no_mask_embed.weight.detach(), fcs=fcs, grad_offsets=offs)
    # no op of a phase writes what another op of the same phase touches (address ranges of the packed table itself)
    assert EMU.check_phase_hazards(prog.fwd_packed) == 0 and EMU.check_phase_hazards(prog.bwd_packed) == 0
    x_in = torch.randn(n, Dh if with_fcs else 256)
    img = torch.randn(n, 256, 256) * 0.5
    d_src, d_hy, d_iou = torch.randn(2, n, 256, 256) * 0.1, torch.randn(n, 32), torch.randn(n)
    src_r, hy_r, iou_r, dx_r = _reference(W, dec, fcs, x_in, img, n, d_src, d_hy, d_iou)

    ws = torch.zeros(prog.fwd_bytes // 4)
    EMU.run(prog.fwd_packed, [0, ws.data_ptr(), 0, x_in.data_ptr(), 0, img.data_ptr(), 0, 0])

    def view(r):
        return ws[r.off // 4: r.off // 4 + r.rows * r.cols].view(r.rows, r.cols)
    assert (view(prog.out["src"]).view(n, 256, 256) - src_r).abs().max() < 2e-5
    assert (view(prog.out["hyper0"]) - hy_r).abs().max() < 2e-5
    assert (view(prog.out["iou4"]) - iou_r).abs().max() < 2e-5

    gflat, wb = torch.zeros(tot), torch.zeros(max(prog.bwd_bytes, 256) // 4)
    for rev in (False, True):            # a phase's ops run concurrently on the device: their order must not matter
        gflat.zero_(); wb.zero_()
        EMU.run(prog.bwd_packed, [0, ws.data_ptr(), wb.data_ptr(), 0, gflat.data_ptr(), d_src.data_ptr(), d_hy.data_ptr(), d_iou.data_ptr()], reverse=rev)
        for p in params:
            g = gflat[offs[id(p)] // 4: offs[id(p)] // 4 + p.numel()].view(p.shape)
            ref = p.grad if p.grad is not None else torch.zeros_like(p)
            # (k_proj biases have a mathematically zero gradient — softmax is invariant to a per-query shift: the floor term covers their noise)
            assert (g - ref).abs().max() <= 1e-4 * ref.abs().max() + 2e-6, (rev, tuple(p.shape))
        r = prog.d_in
        dx = wb[r.off // 4: r.off // 4 + (r.rows - 1) * r.ld + r.cols].as_strided((r.rows, r.cols), (r.ld, 1))
        assert (dx - dx_r).abs().max() <= 1e-4 * dx_r.abs().max()
    # gradients ACCUMULATE into the buffer (gradient accumulation steps): a second backward doubles them
    before = gflat.clone()
    EMU.run(prog.bwd_packed, [0, ws.data_ptr(), wb.data_ptr(), 0, gflat.data_ptr(), d_src.data_ptr(), d_hy.data_ptr(), d_iou.data_ptr()])
    assert (gflat - 2 * before).abs().max() <= 1e-5 * before.abs().max()


def test_frozen_families_emit_no_weight_gradient_ops():
    """--sft_modules text_hidden_fcs only: the decoder's weight-gradient GEMMs are not in the program, the fcs gradients are unchanged."""
    n, Dh = 2, 192
    W, dec, pe, fcs, params, offs, tot = _setup(n, Dh, seed=11)
    full = TP.TailProgram(dec, n, pe.dense_pe_tokens(), pe.no_mask_embed.weight.detach(), fcs=fcs, grad_offsets=offs)
    fc_only = {id(p): offs[id(p)] for p in list(fcs[0].parameters()) + list(fcs[1].parameters())}
    part = TP.TailProgram(dec, n, pe.dense_pe_tokens(), pe.no_mask_embed.weight.detach(), fcs=fcs, grad_offsets=fc_only, hidden_grad=False)
    assert len(part.bwd_packed[0]) < len(full.bwd_packed[0]) and part.d_in is None
    x_in, img = torch.randn(n, Dh), torch.randn(n, 256, 256) * 0.5
    d_src, d_hy, d_iou = torch.randn(2, n, 256, 256) * 0.1, torch.randn(n, 32), torch.randn(n)
    outs = []
    for prog in (full, part):
        ws, g, wb = torch.zeros(prog.fwd_bytes // 4), torch.zeros(tot), torch.zeros(max(prog.bwd_bytes, 256) // 4)
        EMU.run(prog.fwd_packed, [0, ws.data_ptr(), 0, x_in.data_ptr(), 0, img.data_ptr(), 0, 0])
        EMU.run(prog.bwd_packed, [0, ws.data_ptr(), wb.data_ptr(), 0, g.data_ptr(), d_src.data_ptr(), d_hy.data_ptr(), d_iou.data_ptr()])
        outs.append(g)
    k = sum(p.numel() for p in list(fcs[0].parameters()) + list(fcs[1].parameters()))
    assert torch.equal(outs[0][:k], outs[1][:k]) and outs[1][k:].abs().max() == 0 and outs[0][k:].abs().max() > 0


def test_op_table_layout():
    """The packed record is the 256-byte struct tail_program.hip declares (field order and offsets)."""
    dt = TP.OP_DTYPE
    assert dt.itemsize == 256
    assert [dt.fields[k][1] for k in ("type", "flags", "ntiles", "tile_begin", "M", "N", "K", "i0", "i1", "i2", "i3", "pad0")] == list(range(0, 48, 4))
    assert dt.fields["f0"][1] == 48 and dt.fields["ld"][1] == 64 and dt.fields["p"][1] == 160
    src = open(TP.__file__.replace("tail_program.py", "csrc/tail_program.hip")).read()
    assert "int type, flags, ntiles, tile_begin;" in src and "int64_t ld[12];" in src and "uint64_t p[12];" in src
    for name, val in (("OP_GEMM", 1), ("OP_REDUCE", 2), ("OP_LN_FWD", 3), ("OP_LN_BWD", 4), ("OP_ATTN_FWD", 5), ("OP_ATTN_BWD", 6), ("OP_COPY2D", 7)):
        assert f"{name} = {val}" in src and getattr(TP, name) == val


def test_fused_upsampler_backward_program():
    """TailProgram(fused_upsampler=True): the backward takes mp_mask_upsample_fused_bwd_bf16's five outputs in one buffer + the bf16 tokens and
    finishes that kernel's work itself.  Interpreter run on random stand-ins for the kernel's outputs against (a) the plain program fed the
    same d src halves / d hyper0 (every transformer / fcs gradient identical) and (b) the host formulas of autograd_ops.FusedUpsampleMaskFn
    for the six upsampler gradients (the weight gradients land in the reference's [Cin, Cout, 2, 2] layout directly)."""
    from medplib_amd import ops as MO
    n, Dh = 2, 192
    W, dec, pe, fcs, params, offs, tot = _setup(n, Dh, seed=17)
    plain = TP.TailProgram(dec, n, pe.dense_pe_tokens(), pe.no_mask_embed.weight.detach(), fcs=fcs, grad_offsets=offs)
    fused = TP.TailProgram(dec, n, pe.dense_pe_tokens(), pe.no_mask_embed.weight.detach(), fcs=fcs, grad_offsets=offs, fused_upsampler=True)
    assert EMU.check_phase_hazards(fused.bwd_packed) == 0
    Tk = 256
    o, total = MO.upsample_bwd_layout(n, Tk)
    ubuf = torch.randn(total) * 0.1
    dx2 = ubuf[o[0]:o[0] + 2 * n * Tk * 256].view(2, n, Tk, 256)
    dy1 = ubuf[o[1]:o[1] + n * Tk * 256].view(n * Tk, 256)
    a1 = ubuf[o[2]:o[2] + n * Tk * 256].view(n * Tk * 4, 64)
    dy2 = ubuf[o[3]:o[3] + n * Tk * 512].view(n * Tk * 4, 128)
    part = ubuf[o[4]:o[4] + (n * Tk // 8) * 256].view(n * Tk // 8, 256)
    src_bf = (torch.randn(n * Tk, 256) * 0.5).to(torch.bfloat16)
    x_in, img, d_iou = torch.randn(n, Dh), torch.randn(n, 256, 256) * 0.5, torch.randn(n)
    d_hy = part[:, 224:].reshape(n, -1, 32).sum(1).contiguous()
    res = []
    for prog in (plain, fused):
        ws, g, wb = torch.zeros(prog.fwd_bytes // 4), torch.zeros(tot), torch.zeros(max(prog.bwd_bytes, 256) // 4)
        EMU.run(prog.fwd_packed, [0, ws.data_ptr(), 0, x_in.data_ptr(), 0, img.data_ptr(), 0, 0])
        if prog is plain:
            EMU.run(prog.bwd_packed, [0, ws.data_ptr(), wb.data_ptr(), 0, g.data_ptr(), dx2.data_ptr(), d_hy.data_ptr(), d_iou.data_ptr()])
        else:
            EMU.run(prog.bwd_packed, [0, ws.data_ptr(), wb.data_ptr(), 0, g.data_ptr(), ubuf.data_ptr(), src_bf.data_ptr(), d_iou.data_ptr()])
        res.append(g)
    up = dec.output_upscaling
    ups = {id(p) for p in (up[0].weight, up[0].bias, up[1].weight, up[1].bias, up[3].weight, up[3].bias)}

    def grad(g, p):
        return g[offs[id(p)] // 4: offs[id(p)] // 4 + p.numel()].view(p.shape)
    for p in params:
        if id(p) not in ups:
            a, b = grad(res[0], p), grad(res[1], p)
            assert (a - b).abs().max() <= 1e-5 * a.abs().max() + 2e-6, tuple(p.shape)     # (floor: the k_proj biases, whose gradient is rounding noise)
    assert all(grad(res[0], p).abs().max() == 0 for p in params if id(p) in ups)         # the plain program leaves them to FusedUpsampleMaskFn
    cs = part.sum(0)
    want = {id(up[0].weight): (dy1.t() @ src_bf.float()).view(2, 2, 64, 256).permute(3, 2, 0, 1), id(up[3].weight): (dy2.t() @ a1).view(2, 2, 32, 64).permute(3, 2, 0, 1),
            id(up[0].bias): cs[0:64], id(up[1].weight): cs[64:128], id(up[1].bias): cs[128:192], id(up[3].bias): cs[192:224]}
    for p in params:
        if id(p) in ups:
            got, ref = grad(res[1], p), want[id(p)]
            assert (got - ref).abs().max() <= 2e-5 * ref.abs().max(), tuple(p.shape)
