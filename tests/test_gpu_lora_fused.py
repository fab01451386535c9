"""Round 5 fusions of the LoRA training step (dense LlamaMLP with adapters on gate / up / down, scripts/train_stage3.sh:29-33), each held to
BIT identity with the kernels it replaces."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from medplib_amd import ops   # noqa: E402


@pytest.mark.parametrize("T,ff,R,p", [(100, 320, 8, 0.0), (777, 11008, 8, 0.05), (64, 1024, 16, 0.1), (33, 544, 32, 0.05)])
def test_lora_up_add_swiglu_bwd_is_the_two_kernels_in_one_pass(T, ff, R, p):
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(T + ff)
    dt = (torch.randn(T, 64, generator=g, device=dev) * 0.1).to(torch.bfloat16)
    AT = (torch.randn(ff, 64, generator=g, device=dev) * 0.05).to(torch.bfloat16)
    dact = torch.randn(T, ff, generator=g, device=dev).to(torch.bfloat16)
    gu = torch.randn(T, 2 * ff, generator=g, device=dev).to(torch.bfloat16)
    seed = 1234567
    ref = ops.swiglu_pair_bwd(gu, ops.lora_up_add(dt, AT, dact.clone(), R, p, seed))
    keep = dact.clone()
    got = ops.lora_up_add_swiglu_bwd(dt, AT, dact, gu, R, p, seed)
    torch.cuda.synchronize()
    assert torch.equal(dact, keep)                                   # the input gradient itself is left alone
    assert torch.equal(got.view(torch.int16), ref.view(torch.int16)), f"{(got != ref).sum().item()} of {got.numel()} values differ"


@pytest.mark.parametrize("T,K,R,p", [(300, 1024, 8, 0.05), (777, 2816, 16, 0.1), (65, 512, 32, 0.5)])
def test_lora_dropout_mask_bytes_replace_the_regenerated_mask(T, K, R, p):
    """Round 5: mp_lora_down_bf16 leaves lora_dropout's mask as bytes [T, K / 8] and the two backward kernels that need the same mask read
    them instead of hashing again.  The bytes ARE mp_dropout_bf16's mask, and each kernel gives the same bits with them as without."""
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(T + K)
    seed = 424242 + R
    x = torch.randn(T, K, generator=g, device=dev).to(torch.bfloat16)
    A = torch.zeros(64, K, dtype=torch.bfloat16, device=dev)
    A[:R] = (torch.randn(R, K, generator=g, device=dev) * 0.05).to(torch.bfloat16)
    t0 = torch.empty((T, 64), dtype=torch.bfloat16, device=dev); t1 = torch.empty_like(t0)
    kb = ops.keep_bits_for(x)
    assert kb is not None and kb.shape == (T, K // 8)
    ops.lora_down(x, A, t0, R, p, seed)
    ops.lora_down(x, A, t1, R, p, seed, keep_bits=kb)
    assert torch.equal(t0, t1)
    # the bytes against the materialised mask (ones through mp_dropout_bf16: kept elements are 1 / (1 - p), dropped ones 0)
    kept = ops.dropout_bf16(torch.ones(T, K, dtype=torch.bfloat16, device=dev), p, seed) != 0
    bits = ((kb.view(T, K // 8, 1).to(torch.int32) >> torch.arange(8, device=dev, dtype=torch.int32)) & 1).bool().view(T, K)
    assert torch.equal(bits, kept)
    assert abs(kept.float().mean().item() - (1 - p)) < 0.02
    # weight gradient: dropout(x)^T dt
    dt = (torch.randn(T, 64, generator=g, device=dev) * 0.1).to(torch.bfloat16)
    a = ops.tn_skinny(x, dt, R, 1.0, p, seed)
    b = ops.tn_skinny(x, dt, R, 1.0, p, seed, keep_bits=kb)
    assert torch.equal(a, b) and a.abs().max() > 0
    # input gradient: dx + dropout(dt A), alone and with the SwiGLU backward behind it
    if R <= 32:
        AT = A.t().contiguous()
        dx = torch.randn(T, K, generator=g, device=dev).to(torch.bfloat16)
        a = ops.lora_up_add(dt, AT, dx.clone(), R, p, seed)
        b = ops.lora_up_add(dt, AT, dx.clone(), R, p, seed, keep_bits=kb)
        assert torch.equal(a, b) and not torch.equal(a, dx)
        if K % 32 == 0:
            gu = torch.randn(T, 2 * K, generator=g, device=dev).to(torch.bfloat16)
            a = ops.lora_up_add_swiglu_bwd(dt, AT, dx, gu, R, p, seed)
            b = ops.lora_up_add_swiglu_bwd(dt, AT, dx, gu, R, p, seed, keep_bits=kb)
            assert torch.equal(a, b)


@pytest.mark.parametrize("T,N,R", [(300, 1024, 8), (5112, 4096, 8), (777, 2816, 16), (1000, 22016, 16), (65, 520, 32)])
def test_tn_skinny_down_is_the_two_kernels_in_one_pass(T, N, R):
    """Round 5: dB = dY^T t and dt = dY B from one pass over dY (mp_tn_skinny_down_f32).  dB: the bits of mp_tn_skinny_f32 (reduced and as chunk
    partials); dt: mp_lora_down_bf16's value with the fp32 partial sums taken over 256-column blocks instead of eight K ranges — equal to
    the rounding of the bf16 result, and the same bits on every launch."""
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(T + N + R)
    dy = torch.randn(T, N, generator=g, device=dev).to(torch.bfloat16)
    t = (torch.randn(T, 64, generator=g, device=dev) * 0.1).to(torch.bfloat16)
    Bt = torch.zeros(64, N, dtype=torch.bfloat16, device=dev)
    Bt[:R] = (torch.randn(R, N, generator=g, device=dev) * 0.05).to(torch.bfloat16)
    ref_dB = ops.tn_skinny(dy, t, R, 2.0)
    dB, dt = ops.tn_skinny_down(dy, t, Bt, R, 2.0, 2.0)
    assert torch.equal(dB, ref_dB)
    part, dt2 = ops.tn_skinny_down(dy, t, Bt, R, 2.0, 2.0, reduce=False)
    assert torch.equal(part.partial, ops.tn_skinny(dy, t, R, 2.0, reduce=False).partial) and torch.equal(dt, dt2)
    want = 2.0 * (dy.float() @ Bt[:R].float().t())
    assert not dt[:, R:].any()                                       # zero beyond the R rank columns, up to the 64 of the padded tensor
    err = (dt[:, :R].float() - want).abs().max().item()
    assert err <= 2 ** -8 * want.abs().max().item() + 1e-6, (err, want.abs().max().item())
    if N % 256 == 0:
        old = ops.lora_down(dy, Bt, torch.empty((T, 64), dtype=torch.bfloat16, device=dev), R, alpha=2.0)
        d = (old[:, :R].float() - dt[:, :R].float()).abs().max().item()
        assert d <= 2 ** -7 * want.abs().max().item(), d              # one bf16 ulp of the largest entry: different fp32 summation order only


@pytest.mark.parametrize("T,R,p,with_add", [(100, 16, 0.0, True), (1300, 16, 0.05, True), (777, 8, 0.1, False), (5112, 16, 0.05, True), (1, 8, 0.0, False)])
def test_rmsnorm_bwd_with_the_adapter_term_folded_in_is_bit_identical(T, R, p, with_add):
    """Round 5: mp_rmsnorm_bwd_up_bf16 = mp_lora_up_add_bf16 followed by mp_rmsnorm_bwd_bf16 (the gate|up adapter's input gradient has one reader),
    same bits — with the mask regenerated from the seed and with the forward's mask bytes."""
    dev = torch.device("cuda:0")
    d = 4096
    g = torch.Generator(device=dev).manual_seed(T + R)
    x = torch.randn(T, d, generator=g, device=dev).to(torch.bfloat16)
    w = 1.0 + 0.1 * torch.randn(d, generator=g, device=dev)
    dy = torch.randn(T, d, generator=g, device=dev).to(torch.bfloat16)
    add = torch.randn(T, d, generator=g, device=dev).to(torch.bfloat16) if with_add else None
    dt = (torch.randn(T, 64, generator=g, device=dev) * 0.3).to(torch.bfloat16)
    AT = torch.zeros(d, 64, dtype=torch.bfloat16, device=dev)
    AT[:, :R] = (torch.randn(d, R, generator=g, device=dev) * 0.2).to(torch.bfloat16)
    seed = 99 + T
    keep = dy.clone()
    ref = ops.rmsnorm_bwd(x, w, ops.lora_up_add(dt, AT, dy.clone(), R, p, seed), 1e-5, add=add)
    got = ops.rmsnorm_bwd_up(x, w, dy, 1e-5, dt, AT, R, p, seed, add=add)
    torch.cuda.synchronize()
    assert torch.equal(dy, keep)
    assert torch.equal(got.view(torch.int16), ref.view(torch.int16)), f"{(got != ref).sum().item()} of {got.numel()} values differ"
    assert not torch.equal(ref, ops.rmsnorm_bwd(x, w, dy, 1e-5, add=add))            # the adapter term is not a no-op in this test
    if p > 0:
        kb = ops.keep_bits_for(x)
        A = AT.t().contiguous()
        ops.lora_down(x, A, torch.empty((T, 64), dtype=torch.bfloat16, device=dev), R, p, seed, keep_bits=kb)     # the forward leaves the mask bytes
        got2 = ops.rmsnorm_bwd_up(x, w, dy, 1e-5, dt, AT, R, p, seed, add=add, keep_bits=kb)
        assert torch.equal(got2.view(torch.int16), ref.view(torch.int16))


@pytest.mark.parametrize("T,ff,Rd,Rg,p", [(300, 512, 8, 16, 0.0), (1000, 11008, 8, 16, 0.05), (65, 544, 16, 32, 0.1), (777, 1024, 8, 8, 0.05)])
def test_swiglu_bwd_and_both_gate_up_adapter_products_in_one_kernel(T, ff, Rd, Rg, p):
    """Round 5: mp_swiglu_bwd_skinny_f32 = mp_lora_up_add_swiglu_bwd_bf16 (d gate|up from d_act, the down adapter's term and gate|up) followed
    by mp_tn_skinny_down_f32 on its result (the gate|up adapter's dB and dt) — d gate|up is produced in the tile the products read: same bits."""
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(T + ff)
    dact = torch.randn(T, ff, generator=g, device=dev).to(torch.bfloat16)
    gu = torch.randn(T, 2 * ff, generator=g, device=dev).to(torch.bfloat16)
    dtd = (torch.randn(T, 64, generator=g, device=dev) * 0.1).to(torch.bfloat16)
    ATd = torch.zeros(ff, 64, dtype=torch.bfloat16, device=dev); ATd[:, :Rd] = (torch.randn(ff, Rd, generator=g, device=dev) * 0.05).to(torch.bfloat16)
    tg = (torch.randn(T, 64, generator=g, device=dev) * 0.1).to(torch.bfloat16)
    Bt = torch.zeros(64, 2 * ff, dtype=torch.bfloat16, device=dev); Bt[:Rg] = (torch.randn(Rg, 2 * ff, generator=g, device=dev) * 0.05).to(torch.bfloat16)
    seed = 31337
    ref_dgu = ops.lora_up_add_swiglu_bwd(dtd, ATd, dact, gu, Rd, p, seed)
    ref_dB, ref_dt = ops.tn_skinny_down(ref_dgu, tg, Bt, Rg, 2.0, 2.0)
    dgu, dB, dt = ops.swiglu_bwd_skinny(dtd, ATd, dact, gu, Rd, p, seed, tg, Bt, Rg, 2.0, 2.0)
    torch.cuda.synchronize()
    assert torch.equal(dgu.view(torch.int16), ref_dgu.view(torch.int16)), f"{(dgu != ref_dgu).sum().item()} of {dgu.numel()} d gate|up values differ"
    assert torch.equal(dB, ref_dB) and torch.equal(dt, ref_dt) and dB.abs().max() > 0 and dt.float().abs().max() > 0
    if p > 0 and ff % 256 == 0:
        kb = ops.keep_bits_for(dact)
        ops.lora_down(dact, ATd.t().contiguous(), torch.empty((T, 64), dtype=torch.bfloat16, device=dev), Rd, p, seed, keep_bits=kb)
        dgu2, dB2, dt2 = ops.swiglu_bwd_skinny(dtd, ATd, dact, gu, Rd, p, seed, tg, Bt, Rg, 2.0, 2.0, keep_bits=kb)
        assert torch.equal(dgu2.view(torch.int16), ref_dgu.view(torch.int16)) and torch.equal(dB2, ref_dB) and torch.equal(dt2, ref_dt)


@pytest.mark.parametrize("T,fin,fout,r,k0,R", [(5112, 4096, 11008, 8, 8, 16), (600, 1024, 512, 8, 0, 8), (300, 512, 768, 16, 16, 32), (257, 520, 264, 6, 0, 8)])
def test_lora_grad_unpack_from_chunk_partials(T, fin, fout, r, k0, R):
    """mp_lora_grad_unpack_partials_f32: gB += scaleB * sum_c dBp[c][rows[o], k0 + j], gA += scaleA * sum_c dATp[c][col, k0 + i], chunks ascending.
    (A four-ranks-per-thread form of the kernel — 16-byte loads instead of one float per 64-byte line for gA — was measured in round 5: the
    LoRA step did not move, 123.9 against 124.0 ms, and it was not kept.)"""
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(T + fin)
    W = fout + 40                                                    # rows of the fused group's padded B
    rows = torch.randperm(W, generator=g, device=dev)[:fout].sort().values
    x = torch.randn(T, W, generator=g, device=dev).to(torch.bfloat16)
    t = torch.randn(T, 64, generator=g, device=dev).to(torch.bfloat16)
    xin = torch.randn(T, fin, generator=g, device=dev).to(torch.bfloat16)
    dB = ops.tn_skinny(x, t, R, 1.5, reduce=False)
    dAT = ops.tn_skinny(xin, t, R, 1.0, reduce=False)
    gB = torch.randn(fout, r, generator=g, device=dev); gA = torch.randn(r, fin, generator=g, device=dev)
    wantB, wantA = gB.clone(), gA.clone()
    sB = torch.zeros(W * R, device=dev); sA = torch.zeros(fin * R, device=dev)
    for c in range(dB.chunks):
        sB = sB + dB.partial.view(dB.chunks, -1)[c]; sA = sA + dAT.partial.view(dAT.chunks, -1)[c]
    wantB += 1.5 * sB.view(W, R)[rows][:, k0:k0 + r]
    wantA += sA.view(fin, R)[:, k0:k0 + r].t()
    ops.lora_grad_unpack_partials(dB, dAT, rows, k0, gB, gA)
    torch.cuda.synchronize()
    for got, want in ((gB, wantB), (gA, wantA)):
        assert (got - want).abs().max().item() <= 1e-6 * want.abs().max().item() + 1e-7      # (a fused multiply-add against a multiply and an add)
