"""Size-independent properties of the decoder stack AT THE BENCHMARK'S DIMENSIONS (d = 4096, ff = 11008, 32 heads x 128, B = 8, S = 639: the
5112-row launches bench.py times), where the CPU oracle needs minutes per layer: what the reference's decoder guarantees by construction
(HF LlamaModel + DeepSpeed MoE, model/medplib/model/language_model/medplib_moe_llama.py:104-148) and every tiling / fusion decision of the HIP path
has to preserve —
  * a sample's hidden states do not depend on its position in the batch: bit for bit as long as its rows stay on the same side of a launch's
    K-split tail region (the one schedule fact a row's rounding depends on), to fp32 summation order across it,
  * causality: the hidden states of positions < t do not depend on the tokens at positions >= t,
  * right-padding (attention_mask 0 on a suffix) leaves the valid positions' hidden states untouched,
  * the top-1 gate is a per-token function: a token's expert does not depend on where the token sits.
The full-size comparisons against the oracle itself are tests/test_gpu_model.py (8 layers) and bench.py's parity leg (32 layers)."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from medplib_amd.model.config import MedPLIBConfig        # noqa: E402
from oracle import model as OM                      # noqa: E402


def _two_layer_weights(cfg):
    """Two DISTINCT seeded layers at cfg's dims in the HF key layout (oracle.model.init_decoder_layer_weights builds one)."""
    W, g = OM.init_decoder_layer_weights(cfg, seed=5)
    W1, _ = OM.init_decoder_layer_weights(cfg, seed=6)
    for k, v in W1.items():
        if k.startswith("model.layers.0."):
            W["model.layers.1." + k[len("model.layers.0."):]] = v
    return W, g


def _stack(cfg, dev, W):
    from medplib_amd.model.llama import LlamaStack
    llm = LlamaStack(cfg, dev)
    llm.load_hf(W)
    return llm


@pytest.mark.parametrize("moe", [False, True])
def test_decoder_properties_at_the_benchmark_dims(dev, moe):
    cfg = MedPLIBConfig.medplib_7b(num_hidden_layers=2, vocab_size=1024, moe_enable=moe, moe_gate_sampling=False)
    W, g = _two_layer_weights(cfg)
    llm = _stack(cfg, dev, W)
    B, S = 8, 639
    emb = (torch.randn(B, S, cfg.hidden_size, generator=g) * 0.5).to(torch.bfloat16).to(dev)
    kv = torch.ones(B, S, dtype=torch.uint8, device=dev)
    base, _, rt0 = llm.forward(emb, kv, collect_routing=True)
    again, _, _ = llm.forward(emb, kv, collect_routing=True)
    assert torch.equal(base, again), "the same launch twice: the stack is run-to-run deterministic"

    # ---- batch permutation.  A row's bits depend on its operands and on ONE schedule fact: whether its output tile is a whole-K tile or one of
    # the K-split tiles of a launch's last partial wave (csrc/gemm320_bf16.hip: the dense gate|up GEMM at 5112 rows is 1376 tiles = 5 waves + 96
    # tail tiles cut in two along K: rows >= 3840 x the last 24 column tiles) — the two K halves are added in a fixed order, so the result is
    # reproducible, but it is another association of the same sum.  Hence: samples that stay inside (or outside) that region keep their bits
    # wherever they move; samples that cross it agree to fp32 summation order.
    inner = torch.tensor([3, 0, 4, 1, 2, 5, 6, 7], device=dev)        # shuffles positions 0-4 (rows < 3195), the last three stay put
    outi, _, rti = llm.forward(emb[inner].contiguous(), kv[inner].contiguous(), collect_routing=True)
    perm = torch.tensor([3, 0, 7, 1, 6, 2, 5, 4], device=dev)         # moves samples across the region's edge
    outp, _, rtp = llm.forward(emb[perm].contiguous(), kv[perm].contiguous(), collect_routing=True)
    if not moe:
        assert torch.equal(outi, base[inner]), "dense stack: a sample's hidden states moved with its batch position"
        lower = (perm < 4).nonzero().flatten()                       # old positions 0-3 land on new positions 0, 1, 3, 5: outside the region both times
        assert torch.equal(outp[lower], base[perm][lower])
        a, b = outp.float(), base[perm].float()
        err, scale = (a - b).abs().max().item(), b.abs().max().item()
        rows = ((a - b).abs() > 0).any(-1)
        print(f"dense permutation across the K-split region: {int(rows.sum())} of {B * S} rows differ, worst {err:.3e} of {scale:.3f}")
        assert err <= 2 ** -6 * scale, (err, scale)
    else:
        # the gate is row-wise: every token keeps its expert in the first MoE layer (the second layer's input is only equal up to the summation
        # order below).  The expert GEMMs run on routed slabs: a token's slab row — and with it whether its tile is a whole-K tile or a K-split
        # tail tile — follows the token order, so the hidden states agree to fp32 summation order, not bit for bit.
        e0 = rt0[0][0].view(B, S)
        for pm, rt in ((inner, rti), (perm, rtp)):
            assert torch.equal(rt[0][0].view(B, S), e0[pm]), "a token's expert changed with its batch position"
        kept0, keptp = (rt0[0][1].view(B, S) >= 0), (rtp[0][1].view(B, S) >= 0)
        both = (keptp & kept0[perm]) & ((rtp[1][1].view(B, S) >= 0) & (rt0[1][1].view(B, S) >= 0)[perm]) & (rtp[1][0].view(B, S) == rt0[1][0].view(B, S)[perm])
        a, b = outp.float()[both], base[perm].float()[both]
        err = (a - b).abs().max().item()
        scale = b.abs().max().item()
        print(f"MoE permutation: {int(both.sum())} of {B * S} rows comparable, worst difference {err:.3e} of {scale:.3f}")
        assert both.float().mean().item() > 0.95
        assert err <= 2 ** -6 * scale, (err, scale)                 # a few bf16 roundings of O(1) values that saw another summation order

    # ---- causality: new tokens at positions >= t leave positions < t alone
    t = 400
    emb2 = emb.clone()
    emb2[:, t:] = (torch.randn(B, S - t, cfg.hidden_size, generator=torch.Generator().manual_seed(9)) * 0.5).to(torch.bfloat16).to(dev)
    out2, _, rt2 = llm.forward(emb2, kv, collect_routing=True)
    if not moe:
        assert torch.equal(out2[:, :t], base[:, :t]), "dense stack: positions < t changed with the tokens at >= t"
    else:
        same = torch.ones(B, S, dtype=torch.bool, device=dev)
        for l in range(2):
            same &= (rt2[l][0].view(B, S) == rt0[l][0].view(B, S)) & (rt2[l][1].view(B, S) >= 0) & (rt0[l][1].view(B, S) >= 0)
        same = same[:, :t]
        a, b = out2[:, :t].float()[same], base[:, :t].float()[same]
        err, scale = (a - b).abs().max().item(), b.abs().max().item()
        print(f"MoE causality: {int(same.sum())} of {B * t} prefix rows comparable, worst difference {err:.3e} of {scale:.3f}")
        assert torch.equal(rt2[0][0].view(B, S)[:, :t], rt0[0][0].view(B, S)[:, :t]), "a prefix token's first-layer expert changed with the suffix"
        assert same.float().mean().item() > 0.95 and err <= 2 ** -6 * scale, (err, scale)

    # ---- right padding: attention_mask 0 on a suffix of some samples leaves their valid positions alone (dense: bit for bit)
    kv3 = kv.clone()
    kv3[1, 500:] = 0; kv3[4, 17:] = 0; kv3[6, 638:] = 0
    out3, _, rt3 = llm.forward(emb, kv3, collect_routing=True)
    valid = kv3.bool()
    if not moe:
        assert torch.equal(out3[valid], base[valid]), "dense stack: valid positions changed under right padding"
    else:
        assert torch.equal(rt3[0][0].view(B, S)[valid], rt0[0][0].view(B, S)[valid])
        same = valid.clone()
        for l in range(2):
            same &= (rt3[l][0].view(B, S) == rt0[l][0].view(B, S)) & (rt3[l][1].view(B, S) >= 0) & (rt0[l][1].view(B, S) >= 0)
        a, b = out3.float()[same], base.float()[same]
        err, scale = (a - b).abs().max().item(), b.abs().max().item()
        print(f"MoE padding: {int(same.sum())} of {int(valid.sum())} valid rows comparable, worst difference {err:.3e} of {scale:.3f}")
        assert err <= 2 ** -6 * scale, (err, scale)


@pytest.mark.parametrize("moe,targets", [(False, "gate_proj,up_proj,down_proj"), (False, "q_proj,k_proj,v_proj,o_proj,gate_proj,up_proj,down_proj"),
                                         (True, "gate_proj,up_proj,down_proj")])
def test_fresh_adapters_leave_the_forward_unchanged(dev, moe, targets):
    """peft's initialisation (lora_B = 0, peft.tuners.lora.LoraLayer.reset_lora_parameters; the reference attaches its adapters this way,
    train_ds_medplib.py:262-303) makes a freshly wrapped model compute the base model: y = W x + (alpha / r) B A dropout(x) = W x.  Here the adapter
    product rides as 64 extension columns of the frozen projection's K dimension through the training-mode kernels (unfused
    gate|up + SwiGLU pair, kept activations), so the property also says those kernels round where the frozen path's fused ones do: every loss of
    the wrapped model equals the frozen model's bit for bit, with lora_dropout active."""
    from medplib_amd import engine
    from medplib_amd.model.medplib import LISAForCausalLM, MedPLIBForCausalLM
    from oracle import ops as O
    cfg = MedPLIBConfig.tiny(moe_enable=moe, sam_depth=2, num_hidden_layers=3)
    W = OM.init_hf_weights(cfg)
    batch = OM.make_batch(cfg, 3, seed=17, ragged=True)
    gb = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in batch.items()}
    gb["masks_list"] = [x.to(dev) for x in batch["masks_list"]]
    losses = []
    for wrap in (False, True):
        torch.manual_seed(1234)
        m = (MedPLIBForCausalLM if moe else LISAForCausalLM)(cfg, device=dev)
        m.load_hf_state_dict(W)
        m.train()
        if wrap:
            m.enable_lora(lora_r=8, lora_alpha=16, lora_dropout=0.05, lora_target_modules=targets)
        eng, _, _, _ = engine.initialize(model=m, model_parameters=m.trainable_parameters(),
                                         config={"optimizer": {"params": {"lr": 1e-4, "betas": (0.9, 0.95)}}, "gradient_clipping": 1.0})
        out = eng(**gb)
        torch.cuda.synchronize()
        losses.append({k: out[k].detach().float().cpu().clone() for k in O.LOSS_KEYS})
    for k in losses[0]:
        assert torch.equal(losses[0][k], losses[1][k]), (k, losses[0][k].item(), losses[1][k].item())


@pytest.mark.parametrize("h,w,B", [(64, 64, 8), (16, 16, 8), (24, 48, 3)])
def test_fused_upsampler_is_token_local(dev, h, w, B):
    """Both transposed convolutions have kernel = stride = 2 and LayerNorm2d normalises over channels at ONE pixel
    (model/segment_anything_med2d/modeling/mask_decoder.py:53-59, common.py LayerNorm2d): the 4 x 4 output patch of a token is a function of that
    token alone (and of its image's hypernetwork row for the mask).  So for ANY shuffle of the tokens over the grid positions and the images of the
    batch, the fused kernel's output patches must be the same patches, shuffled — bit for bit (whichever wave, workgroup or XCD a token lands on) —
    at the roofline benchmark's own size (64 x 64 tokens, batch 8: the 50.5 MB launch) and at the model's geometry."""
    from medplib_amd import ops
    from oracle import sam as OS
    W = OS.init_weights(seed=7)
    g = torch.Generator().manual_seed(h * 131 + w)
    tok = (torch.randn(B, h * w, 256, generator=g)).to(torch.bfloat16).to(dev)
    hyper = torch.randn(1, 32, generator=g).expand(B, 32).contiguous().to(dev)       # one hypernetwork row for all images: patches may cross images
    w1p, w2p = ops.pack_upsampler_weights(W["mask_decoder.output_upscaling.0.weight"].to(dev), W["mask_decoder.output_upscaling.3.weight"].to(dev))
    d = lambda k: W[k].to(dev)
    tail = (w1p, d("mask_decoder.output_upscaling.0.bias"), d("mask_decoder.output_upscaling.1.weight"), d("mask_decoder.output_upscaling.1.bias"),
            w2p, d("mask_decoder.output_upscaling.3.bias"), h, w)
    up0, mask0 = ops.mask_upsample_fused(tok, *tail, hyper=hyper)
    perm = torch.randperm(B * h * w, generator=g).to(dev)
    tok1 = tok.reshape(B * h * w, 256)[perm].reshape(B, h * w, 256).contiguous()
    up1, mask1 = ops.mask_upsample_fused(tok1, *tail, hyper=hyper)
    torch.cuda.synchronize()

    def patches(up, mask):                       # [B, 32, 4h, 4w], [B, 4h, 4w] -> per token [B h w, 32, 4, 4], [B h w, 4, 4]
        u = up.view(B, 32, h, 4, w, 4).permute(0, 2, 4, 1, 3, 5).reshape(B * h * w, 32, 4, 4)
        m = mask.view(B, h, 4, w, 4).permute(0, 1, 3, 2, 4).reshape(B * h * w, 4, 4)
        return u, m
    u0, m0 = patches(up0, mask0)
    u1, m1 = patches(up1, mask1)
    assert torch.equal(u1, u0[perm]), "an upscaled patch changed with its token's position"
    assert torch.equal(m1, m0[perm]), "a mask patch changed with its token's position"


@pytest.mark.parametrize("moe", [False, True])
def test_decode_is_prefix_consistent_and_equals_the_prefill_at_true_dims(dev, moe):
    """Two properties of `evaluate()`'s KV-cache decode at the 7B layer dims (2 layers, S = 639 after the splice), beyond the 5 tokens the oracle
    comparison (tests/test_gpu_model.py: test_evaluate_at_true_dims) can afford:
    (1) greedy decoding is a deterministic function of the prefix (HF generate with a KV cache, model/MedPLIB.py:574-680): the first n tokens of a
    40-token generation are the n-token generation — across cache lengths that cross a 64-key tile edge of the flash-decoding kernel (640 -> 679),
    two of the graph replay's every-16-token EOS checks, and the device-side cache length (eos_token_id = -1: random weights must not end early);
    (2) a KV cache is an optimisation, not a model change (prepare_inputs_for_generation, medplib_moe_llama.py:451-485): the hidden state the
    decode step computes for generated token t is the hidden state a PREFILL over prompt + generated tokens computes at that position.  Two
    disjoint kernel sets meet here — M = 1 GEMVs (shared / expert-indexed, norm-folded, K-split), RoPE at the device-side position + cache append,
    flash-decoding attention with its split merge, the fused norm + gate + routing launch, HIP-graph replay — against the 320-row GEMMs, the RoPE
    epilogue, the tiled causal attention and the batched expert GEMMs.  Equal to bf16 rounding of O(1) values under another summation order
    (bound: 2^-5 of the largest entry = 4 bf16 ulps there; measured 2 ulps worst, 1 median); a MoE token whose two gate probabilities are within
    that noise may take the other expert in one of the two paths, so there the bound must hold on all but at most two of the rows."""
    from medplib_amd.model.medplib import LISAForCausalLM, MedPLIBForCausalLM
    cfg = MedPLIBConfig.medplib_7b(num_hidden_layers=2, vocab_size=4096, seg_token_idx=4000, moe_enable=moe, moe_gate_sampling=False)
    W = OM.init_hf_weights_aliased(cfg, seed=5)
    m = (MedPLIBForCausalLM if moe else LISAForCausalLM)(cfg, device=dev)
    m.load_hf_state_dict(W)
    m.eval()
    batch = OM.make_batch(cfg, 1, L=64, H=336, Wd=336, seed=11)
    ic = batch["images_clip"].to(torch.bfloat16).float().to(dev)
    ids = batch["input_ids"][:1].numpy()
    n_in = ids.shape[1]
    # ---- (1)
    long, hid = m._greedy(ids, ic, 40, -1)
    long = long[0].tolist()
    assert len(long) == n_in + 40 and long[:n_in] == ids[0].tolist() and len(hid) == 40       # the prompt's states + one per FED token (the last is not fed)
    for n in (1, 12, 33):
        short = m._greedy(ids, ic, n, -1)[0][0].tolist()
        assert short == long[:n_in + n], (n, short[n_in:], long[n_in:n_in + n])
    assert len(set(long[n_in:])) > 1, "a constant generation would make the prefix check vacuous"
    # ---- (2)
    n_new = 20
    S = hid[0].shape[1]
    steps = torch.cat([h.reshape(1, -1) for h in hid[1:n_new]], 0).float()           # [n_new - 1, d]: decode path
    import numpy as np
    _, hid2 = m._greedy(np.asarray([long[:n_in + n_new - 1]], dtype=np.int64), ic, 1, -1)    # prefill over prompt + the 19 fed tokens
    full = hid2[0][0].float()
    assert full.shape[0] == S + n_new - 1
    assert (full[:S] - hid[0][0].float()).abs().max().item() <= 2 ** -6 * full.abs().max().item(), "the prompt's own states moved with the longer prefill"
    pre = full[S:]
    scale = pre.abs().max().item()
    err = (steps - pre).abs().amax(1)
    print(f"moe={moe}: decode vs prefill over {n_new - 1} tokens: worst row error {err.max().item():.3e}, median {err.median().item():.3e}, scale {scale:.3f}")
    bad = int((err > 2 ** -5 * scale).sum())
    assert bad <= (2 if moe else 0), (bad, err.tolist())


@pytest.mark.parametrize("moe,lora", [(True, False), (False, True), (True, True)])
def test_two_equal_micro_steps_are_one_step(dev, moe, lora):
    """DeepSpeed's accumulation window (engine.backward scales the loss by 1 / gradient_accumulation_steps and sums gradients until the
    boundary; the reference trains with --grad_accumulation_steps, train_ds_medplib.py:96,412-419): feeding the SAME micro-batch twice under
    gradient_accumulation_steps = 2 accumulates g / 2 + g / 2 = g exactly (halving and the sum of two equal halves are exact in binary floating
    point), so the optimizer step must leave the parameters of a one-step run bit for bit — which holds only if every gradient producer ADDS into
    the flat buffer in the second micro-step (the tail program's direct gradient stores, the adapters' unpack launches, the fused upsampler's
    backward) and nothing steps, zeroes or re-packs in between."""
    from medplib_amd import engine
    from medplib_amd.model.medplib import LISAForCausalLM, MedPLIBForCausalLM
    cfg = MedPLIBConfig.tiny(moe_enable=moe, sam_depth=2, num_hidden_layers=2)
    W = OM.init_hf_weights(cfg)
    batch = OM.make_batch(cfg, 3, seed=19)
    gb = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in batch.items()}
    gb["masks_list"] = [x.to(dev) for x in batch["masks_list"]]

    def run(gas, micro):
        torch.manual_seed(1234)
        m = (MedPLIBForCausalLM if moe else LISAForCausalLM)(cfg, device=dev)
        m.load_hf_state_dict(W)
        m.train()
        if lora:
            lo = m.enable_lora(lora_r=8, lora_alpha=16, lora_dropout=0.0, lora_target_modules="gate_proj,up_proj,down_proj",
                               sft_modules="mask_decoder,text_hidden_fcs")
            g = torch.Generator().manual_seed(33)
            for n, p_ in zip(lo.names, lo.params):
                if "lora_" in n:
                    p_.data.copy_((torch.randn(p_.shape, generator=g) * 0.04).to(torch.bfloat16).float().to(dev))
        eng, _, _, _ = engine.initialize(model=m, model_parameters=m.trainable_parameters(),
                                         config={"optimizer": {"params": {"lr": 1e-3, "betas": (0.9, 0.95)}}, "gradient_clipping": 1.0,
                                                 "gradient_accumulation_steps": gas})
        snaps = [[p_.detach().float().cpu().clone() for p_ in eng.optimizer.params]]          # [0]: as initialised
        for _ in range(micro):
            out = eng(**gb)
            eng.backward(out["loss"])
            eng.step()
            torch.cuda.synchronize()
            snaps.append([p_.detach().float().cpu().clone() for p_ in eng.optimizer.params])
        return snaps
    init, one = run(1, 1)
    two = run(2, 2)
    assert all(torch.equal(a, b) for a, b in zip(init, two[0])), "the two runs start from different parameters"
    assert any(not torch.equal(a, b) for a, b in zip(init, one)), "the single step changed nothing: the comparison would be vacuous"
    assert all(torch.equal(a, b) for a, b in zip(init, two[1])), "a parameter moved before the accumulation boundary"
    bad = [i for i, (a, b) in enumerate(zip(one, two[2])) if not torch.equal(a, b)]
    assert not bad, f"{len(bad)} of {len(one)} parameters differ between one step and two equal half-steps (first: {bad[:5]})"


@pytest.mark.parametrize("mode", ["moe", "dense_lora"])
def test_identical_ranks_are_one_rank(dev, mode, tmp_path):
    """Data parallelism averages the ranks' gradients (DeepSpeed engine allreduce, train_ds_medplib.py:412-419): two ranks that hold the SAME
    micro-batch reduce g + g and scale by 1 / 2 — both exact — so two optimizer steps must leave the parameters of a one-rank run bit for bit:
    the bucketed SUM all-reduce (one bucket for the frozen trunk, per-layer buckets from inside the decoder backward with adapters), grad_scale
    = 1 / world inside the AdamW kernel and the clipping norm all sit on that path.  Two real processes on cuda:0 over gloo (RCCL refuses two
    ranks on one device; tests/_dp_equiv_worker.py)."""
    import os
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = {}
    for n in (1, 2):
        env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
        env.update({"OMP_NUM_THREADS": "4", "HSA_ENABLE_IPC_MODE_LEGACY": "0"})
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        out = str(tmp_path / f"params_{n}.pt")
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1", "--master-port", str(port),
               os.path.join(root, "tests", "_dp_equiv_worker.py"), out, mode]
        p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
        assert p.returncode == 0, p.stderr[-4000:]
        outs[n] = torch.load(out)
    assert len(outs[1]) == len(outs[2]) and len(outs[1]) > 0
    bad = [i for i, (a, b) in enumerate(zip(outs[1], outs[2])) if not torch.equal(a, b)]
    assert not bad, f"{len(bad)} of {len(outs[1])} parameters differ between one rank and two identical ranks (first: {bad[:5]})"
