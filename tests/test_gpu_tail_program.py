"""The mask-tail program on the MI355X (csrc/tail_program.hip through the C ABI): the persistent kernel against (a) the numpy interpreter of
the SAME op table on the same inputs (oracle/tail_program_emu.py; the lowering itself is pinned to the oracle's autograd on the CPU in
tests/test_tail_program.py), (b) the op-by-op path it replaces, (c) itself across grid sizes and under load (the grid barrier and the
cross-XCD visibility protocol: results must be bit-identical whatever the placement)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from medplib_amd import tail_program as TP                  # noqa: E402
from medplib_amd.model.sam import MaskDecoder, PromptEncoderText   # noqa: E402
from oracle import sam as OS                                 # noqa: E402
from oracle import tail_program_emu as EMU                   # noqa: E402


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _modules(seed, Dh, device):
    torch.manual_seed(seed)
    W = OS.init_weights(seed=seed)
    dec, pe = MaskDecoder(), PromptEncoderText()
    dec.load_state_dict({k[len("mask_decoder."):]: v for k, v in W.items() if k.startswith("mask_decoder.")})
    pe.load_state_dict({k[len("prompt_encoder."):]: v for k, v in W.items() if k.startswith("prompt_encoder.")}, strict=False)
    fc1, fc2 = torch.nn.Linear(Dh, Dh), torch.nn.Linear(Dh, 256)
    return dec.to(device), pe.to(device), fc1.to(device), fc2.to(device)


def _offsets(params):
    offs, tot = {}, 0
    for p in params:
        offs[id(p)] = tot * 4
        tot += -(-p.numel() // 64) * 64
    return offs, tot


@pytest.mark.parametrize("n", [1, 3, 8])
def test_kernel_matches_interpreter(dev, n):
    """Same weights, same inputs, same packed table semantics: device program vs the CPU interpreter, forward outputs and every gradient."""
    Dh = 320
    dec, pe, fc1, fc2 = _modules(5 + n, Dh, dev)
    cdec, cpe, cfc1, cfc2 = _modules(5 + n, Dh, torch.device("cpu"))
    gp = [fc1.weight, fc1.bias, fc2.weight, fc2.bias] + list(dec.parameters())
    cp = [cfc1.weight, cfc1.bias, cfc2.weight, cfc2.bias] + list(cdec.parameters())
    goffs, tot = _offsets(gp)
    coffs, _ = _offsets(cp)
    gprog = TP.TailProgram(dec, n, pe.dense_pe_tokens(), pe.no_mask_embed.weight.detach(), fcs=(fc1, fc2), grad_offsets=goffs)
    cprog = TP.TailProgram(cdec, n, cpe.dense_pe_tokens(), cpe.no_mask_embed.weight.detach(), fcs=(cfc1, cfc2), grad_offsets=coffs)
    assert list(gprog.fwd_packed[2]) == list(cprog.fwd_packed[2]) and list(gprog.bwd_packed[2]) == list(cprog.bwd_packed[2])
    x, img = torch.randn(n, Dh), torch.randn(n, 256, 256) * 0.5
    d_src, d_hy, d_iou = torch.randn(2, n, 256, 256) * 0.1, torch.randn(n, 32), torch.randn(n)
    # CPU
    cws, cg, cwb = torch.zeros(cprog.fwd_bytes // 4), torch.zeros(tot), torch.zeros(max(cprog.bwd_bytes, 256) // 4)
    EMU.run(cprog.fwd_packed, [0, cws.data_ptr(), 0, x.data_ptr(), 0, img.data_ptr(), 0, 0])
    EMU.run(cprog.bwd_packed, [0, cws.data_ptr(), cwb.data_ptr(), 0, cg.data_ptr(), d_src.data_ptr(), d_hy.data_ptr(), d_iou.data_ptr()])
    # device
    ws, src, hy, iou4 = gprog.run_forward(x.to(dev), img.to(dev))
    gg = torch.zeros(tot, device=dev)
    d_in = gprog.run_backward(ws, d_src.to(dev), d_hy.to(dev), d_iou.to(dev), gg.data_ptr())
    torch.cuda.synchronize()
    assert gprog.check_sync()

    def cview(r):
        return cws[r.off // 4: r.off // 4 + r.rows * r.cols].view(r.rows, r.cols)
    for name, got in (("src", src.view(n * 256, 256)), ("hyper0", hy), ("iou4", iou4)):
        ref = cview(cprog.out[name])
        assert (got.cpu() - ref).abs().max() <= 2e-5 * max(1.0, ref.abs().max().item()), name
    worst = 0.0
    for p_g, p_c in zip(gp, cp):
        o, k = goffs[id(p_g)] // 4, p_g.numel()
        a, b = gg[o:o + k].cpu(), cg[coffs[id(p_c)] // 4: coffs[id(p_c)] // 4 + k]
        worst = max(worst, ((a - b).abs().max() / (b.abs().max() + 1e-5)).item())
        assert (a - b).abs().max() <= 2e-4 * b.abs().max() + 5e-6, tuple(p_g.shape)
    r = cprog.d_in
    cdx = cwb[r.off // 4: r.off // 4 + r.rows * r.cols].view(r.rows, r.cols)
    assert (d_in.cpu() - cdx).abs().max() <= 2e-4 * cdx.abs().max()
    print(f"[tail program n={n}] fwd phases {len(gprog.fwd_packed[2])} / ops {len(gprog.fwd_packed[0])}, bwd phases {len(gprog.bwd_packed[2])} / ops "
          f"{len(gprog.bwd_packed[0])}, worst relative gradient difference {worst:.2e}")


def test_bit_identical_across_grids_and_under_load(dev):
    """Every sum has a fixed order and every hand-off goes through the release / acquire barrier: 256, 64 and 7 workgroups, an idle chip and
    one busy with a bandwidth hog on another stream give the same bits, forward and backward, twenty times over."""
    n, Dh = 8, 256
    dec, pe, fc1, fc2 = _modules(21, Dh, dev)
    gp = [fc1.weight, fc1.bias, fc2.weight, fc2.bias] + list(dec.parameters())
    goffs, tot = _offsets(gp)
    prog = TP.TailProgram(dec, n, pe.dense_pe_tokens(), pe.no_mask_embed.weight.detach(), fcs=(fc1, fc2), grad_offsets=goffs)
    x, img = torch.randn(n, Dh, device=dev), torch.randn(n, 256, 256, device=dev) * 0.5
    d_src, d_hy, d_iou = torch.randn(2, n, 256, 256, device=dev) * 0.1, torch.randn(n, 32, device=dev), torch.randn(n, device=dev)
    ref = None
    hog_a, hog_b = torch.randn(64 << 20, device=dev), torch.empty(64 << 20, device=dev)
    side = torch.cuda.Stream()
    for it in range(20):
        grid = (256, 64, 7, 255)[it % 4]
        if it >= 8:                                            # uneven load: copies streaming on another queue while the program runs
            with torch.cuda.stream(side):
                for _ in range(4):
                    hog_b.copy_(hog_a)
        ws, src, hy, iou4 = prog.run_forward(x, img, grid=grid)
        g = torch.zeros(tot, device=dev)
        d_in = prog.run_backward(ws, d_src, d_hy, d_iou, g.data_ptr(), grid=grid)
        cur = [t.clone() for t in (src, hy, iou4, g, d_in)]
        torch.cuda.synchronize()
        assert prog.check_sync()
        if ref is None:
            ref = cur
        else:
            for a, b, nm in zip(cur, ref, ("src", "hyper0", "iou4", "grads", "d_hidden")):
                assert torch.equal(a, b), f"iteration {it} (grid {grid}): {nm} differs in {(a != b).sum().item()} elements"


def test_program_path_matches_op_by_op_path(dev):
    """MaskDecoder.forward with the program against the op-by-op autograd path it replaces (MP_TAIL_PROGRAM=0), fp32 upscaling chain on both:
    masks, iou, parameter gradients and the gradient of the hidden rows."""
    n, Dh = 4, 256
    # (seed: with 7 one ReLU pre-activation of prompt 0 sits within rounding of zero and the two paths' different summation orders put it on
    # different sides — a 1e-3 difference in that prompt's rows against an fp64 evaluation for ONE of the paths, measured in round 5; any
    # seed without such a unit shows both paths within 2e-6 of fp64)
    torch.manual_seed(8)
    x, img = torch.randn(n, Dh, device=dev), torch.randn(n, 256, 256, device=dev) * 0.5
    gl, gi = torch.randn(n, 64, 64, device=dev), torch.randn(n, device=dev)
    res = []
    for use in (True, False):
        dec, pe, fc1, fc2 = _modules(33, Dh, dev)
        dec.use_program = use
        xi = x.clone().requires_grad_()
        low, iou = dec(img, pe.dense_pe_tokens(), pe.no_mask_embed.weight, None, fcs=(fc1, fc2), hidden_rows=xi)
        ((low * gl).sum() + (iou * gi).sum()).backward()
        res.append((low.detach(), iou.detach(), xi.grad, {k: p.grad for k, p in list(dec.named_parameters()) + [("fc1.w", fc1.weight), ("fc1.b", fc1.bias),
                                                                                                                  ("fc2.w", fc2.weight), ("fc2.b", fc2.bias)]}))
    (l0, i0, dx0, g0), (l1, i1, dx1, g1) = res
    assert (l0 - l1).abs().max() <= 2e-5 * l1.abs().max() and (i0 - i1).abs().max() <= 2e-5
    assert (dx0 - dx1).abs().max() <= 2e-4 * dx1.abs().max()
    for k in g1:
        if g1[k] is None:
            assert g0[k] is None or g0[k].abs().max() == 0, k
            continue
        assert (g0[k] - g1[k]).abs().max() <= 2e-4 * g1[k].abs().max() + 5e-6, k


def test_direct_accumulation_into_preset_grads(dev):
    """With every trainable tensor's .grad preset inside ONE buffer (the engine's flat gradient buffer) the backward writes there itself and
    hands autograd None: same numbers as the returned-gradient mode, accumulated over two backward passes."""
    n, Dh = 2, 192
    x, img = torch.randn(n, Dh, device=dev), torch.randn(n, 256, 256, device=dev) * 0.5
    gl, gi = torch.randn(n, 64, 64, device=dev), torch.randn(n, device=dev)
    out = []
    for direct in (False, True):
        dec, pe, fc1, fc2 = _modules(41, Dh, dev)
        ps = list(fc1.parameters()) + list(fc2.parameters()) + list(dec.parameters())
        if direct:
            flat = torch.zeros(sum(p.numel() for p in ps), device=dev)
            o = 0
            for p in ps:
                p.grad = flat[o:o + p.numel()].view(p.shape)
                o += p.numel()
        for _ in range(2):
            low, iou = dec(img, pe.dense_pe_tokens(), pe.no_mask_embed.weight, None, fcs=(fc1, fc2), hidden_rows=x)
            ((low * gl).sum() + (iou * gi).sum()).backward()
        out.append([p.grad.clone() if p.grad is not None else None for p in ps])
    for a, b in zip(*out):
        if a is None:
            assert b is None or b.abs().max() == 0
        else:
            assert (a - b).abs().max() <= 1e-5 * a.abs().max() + 1e-7


def test_fused_upsampler_node_matches_two_node_path(dev):
    """Training through the fused bf16 upsampler: ONE autograd node (program + upsampler kernels; the backward program also finishes the
    upsampler's weight / bias / LayerNorm2d gradients) against the program and FusedUpsampleMaskFn as two nodes: same kernels forward (bit-equal
    masks), every gradient within fp32 summation-order noise."""
    from medplib_amd.model import sam as SAM
    n, Dh = 8, 256
    torch.manual_seed(5)
    x, img = torch.randn(n, Dh, device=dev), torch.randn(n, 256, 256, device=dev) * 0.5
    gl, gi = torch.randn(n, 64, 64, device=dev), torch.randn(n, device=dev)
    res = []
    keep = SAM._FUSED_TAIL
    try:
        for fused in (True, False):
            SAM._FUSED_TAIL = fused
            dec, pe, fc1, fc2 = _modules(51, Dh, dev)
            dec.fused_bf16_upsampler = True
            xi = x.clone().requires_grad_()
            low, iou = dec(img, pe.dense_pe_tokens(), pe.no_mask_embed.weight, None, fcs=(fc1, fc2), hidden_rows=xi)
            ((low * gl).sum() + (iou * gi).sum()).backward()
            named = list(dec.named_parameters()) + [("fc1.w", fc1.weight), ("fc1.b", fc1.bias), ("fc2.w", fc2.weight), ("fc2.b", fc2.bias)]
            res.append((low.detach(), iou.detach(), xi.grad, {k: p.grad for k, p in named}))
    finally:
        SAM._FUSED_TAIL = keep
    (l0, i0, dx0, g0), (l1, i1, dx1, g1) = res
    assert torch.equal(l0, l1) and torch.equal(i0, i1)
    assert (dx0 - dx1).abs().max() <= 1e-5 * dx1.abs().max()
    worst = {}
    for k in g1:
        if g1[k] is None:
            assert g0[k] is None or g0[k].abs().max() == 0, k
            continue
        worst[k] = ((g0[k] - g1[k]).abs().max() / (g1[k].abs().max() + 1e-12)).item()
    # (k_proj biases: their gradient is mathematically zero — softmax is invariant to a per-query shift — so both sides hold rounding noise)
    bad = {k: v for k, v in worst.items() if v > 2e-5 and (g0[k] - g1[k]).abs().max() > 2e-6 and not k.endswith("k_proj.bias")}
    assert not bad, bad


def test_give_up_is_reported_at_the_next_launch_and_switches_to_the_op_by_op_tail(dev):
    """Round-5 advisor: a barrier that cannot complete no longer traps the launch (which poisons the context); every workgroup leaves, sync[1] is set, the
    word comes back through a pinned copy and the NEXT launch of the program raises after switching the decoder to the op-by-op tail.  A barrier that
    really gives up needs tens of seconds of a non-resident workgroup, so the test plants the flag in the read-back word of a finished launch; it also
    checks the normal case (flag clear: nothing raised, the per-launch sync words are distinct allocations) and that gradients dropped between forward and
    backward (zero_grad(set_to_none=True)) do not make the direct form write through stale addresses."""
    n, Dh = 2, 256
    dec, pe, fc1, fc2 = _modules(35, Dh, dev)
    x = torch.randn(n, Dh, device=dev, requires_grad=True)
    img = torch.randn(n, 256, 256, device=dev) * 0.5
    low, iou = dec(img, pe.dense_pe_tokens(), pe.no_mask_embed.weight, None, fcs=(fc1, fc2), hidden_rows=x)
    (low.sum() + iou.sum()).backward()
    torch.cuda.synchronize()
    runner = dec._runner
    prog = next(iter(runner._cache.values()))
    assert prog.check_sync()                                   # the launches so far completed their barriers
    low2, _ = dec(img, pe.dense_pe_tokens(), pe.no_mask_embed.weight, None, fcs=(fc1, fc2), hidden_rows=x)     # flag clear: nothing raised
    torch.cuda.synchronize()
    assert torch.equal(low2, low)
    # plant a give-up in the word the last launch's read-back filled
    assert prog._flag_host is not None
    prog._flag_host[0] = 1
    ev = torch.cuda.Event(); ev.record(); torch.cuda.synchronize()
    prog._pending = ev
    with pytest.raises(RuntimeError, match="gave up at a grid barrier"):
        dec(img, pe.dense_pe_tokens(), pe.no_mask_embed.weight, None, fcs=(fc1, fc2), hidden_rows=x)
    assert dec.use_program is False
    low3, _ = dec(img, pe.dense_pe_tokens(), pe.no_mask_embed.weight, None, fcs=(fc1, fc2), hidden_rows=x)     # the op-by-op tail takes over
    assert (low3 - low).abs().max() <= 2e-5 * low.abs().max()
    # direct accumulation with the gradients dropped between forward and backward
    dec.use_program = True
    params = [p for p in list(dec.parameters()) + list(fc1.parameters()) + list(fc2.parameters()) if p.requires_grad]
    flat = torch.zeros(sum(-(-p.numel() // 64) * 64 for p in params), device=dev)
    off = 0
    for p in params:
        p.grad = flat[off:off + p.numel()].view(p.shape)
        off += -(-p.numel() // 64) * 64
    xa = x.detach().clone().requires_grad_()
    low4, iou4 = dec(img, pe.dense_pe_tokens(), pe.no_mask_embed.weight, None, fcs=(fc1, fc2), hidden_rows=xa)
    for p in params:
        p.grad = None                                          # zero_grad(set_to_none=True) between forward and backward
    (low4.sum() + iou4.sum()).backward()
    torch.cuda.synchronize()
    assert float(flat.abs().max()) == 0.0                       # nothing was written through the addresses taken at forward time
    got = [p.grad for p in params]
    assert all(g is not None for g in got if g is not None) and sum(g is not None for g in got) > 10
    # ... and the gradients equal the ones of the first (gflat) backward of the same inputs
    xb = x.detach().clone().requires_grad_()
    ref_params = {id(p): p.grad.clone() for p in params if p.grad is not None}
    for p in params:
        p.grad = None
    low5, iou5 = dec(img, pe.dense_pe_tokens(), pe.no_mask_embed.weight, None, fcs=(fc1, fc2), hidden_rows=xb)
    (low5.sum() + iou5.sum()).backward()
    for p in params:
        if p.grad is not None:
            assert (p.grad - ref_params[id(p)]).abs().max() <= 1e-5 * ref_params[id(p)].abs().max() + 1e-7
    assert (xa.grad - xb.grad).abs().max() <= 1e-5 * xb.grad.abs().max() + 1e-7
