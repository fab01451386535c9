"""The last decoder layer's MLP runs on the rows something reads (the supervised rows of the filtered CE and the <SEG> rows; DESIGN
section 4): with the switch on and off, every loss and every trainable gradient is the same — bit for bit where the kernels see the same row
arithmetic (the frozen MoE trunk: expert GEMM rows are independent of their slab position), to fp32 summation order where a sum over rows
changes its length (the adapters' weight gradients)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from medplib_amd.model.config import MedPLIBConfig        # noqa: E402
from oracle import model as OM                      # noqa: E402
from oracle import ops as O                         # noqa: E402


def _run(dev, cfg, W, batch, prune, lora_kw=None, adapter_init=None):
    from medplib_amd import engine
    from medplib_amd.model import llama_lora, medplib
    from medplib_amd.model.medplib import LISAForCausalLM, MedPLIBForCausalLM
    medplib._PRUNE_LAST_MLP = prune
    llama_lora._PRUNE_ROWS = prune
    try:
        m = (MedPLIBForCausalLM if cfg.moe_enable else LISAForCausalLM)(cfg, device=dev)
        m.load_hf_state_dict(W)
        m.train()
        if lora_kw is not None:
            lora = m.enable_lora(**lora_kw)
            for p_, v in zip(lora.params, adapter_init(lora)):
                p_.data.copy_(v.to(dev))
        eng, _, _, _ = engine.initialize(model=m, model_parameters=m.trainable_parameters(),
                                         config={"optimizer": {"params": {"lr": 1e-4, "betas": (0.9, 0.95)}}, "gradient_clipping": 1.0})
        gb = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in batch.items()}
        gb["masks_list"] = [x.to(dev) for x in batch["masks_list"]]
        out = eng(**gb)
        active = m.last_pruned is not None
        n_rows = m.last_pruned[0] if active else 0
        assert m.model.llm.needed_rows is None                       # the row set does not outlive the pass it was made for
        eng.backward(out["loss"])
        torch.cuda.synchronize()
        losses = {k: out[k].detach().float().cpu().clone() for k in O.LOSS_KEYS}
        grads = {n: p_.grad.detach().float().cpu().clone() for n, p_ in m.named_parameters() if p_.requires_grad and p_.grad is not None}
        if lora_kw is not None:
            grads.update({n: p_.grad.detach().float().cpu().clone() for n, p_ in zip(m.model.llm.lora.names, m.model.llm.lora.params)})
        return losses, grads, active, n_rows
    finally:
        medplib._PRUNE_LAST_MLP = True
        llama_lora._PRUNE_ROWS = True


def _long_batch(cfg, B, seed, ragged=False):
    """make_batch with most of the answer tokens unsupervised, so the read rows are a minority of the sequence (as in the real data:
    ~1 % of 5112 rows at the benchmark's shapes)."""
    batch = OM.make_batch(cfg, B, seed=seed, ragged=ragged)
    lab = batch["labels"].clone()
    for b in range(B):
        sup = (lab[b] != -100).nonzero().flatten()
        if sup.numel() > 3:
            lab[b, sup[3:]] = -100                       # three supervised tokens per sample (+ the <SEG> rows)
    batch["labels"] = lab
    return batch


@pytest.mark.parametrize("moe,ragged", [(True, False), (False, False), (True, True)])
def test_frozen_trunk_last_layer_mlp_on_read_rows_is_bit_identical(dev, moe, ragged):
    cfg = MedPLIBConfig.tiny(moe_enable=moe, sam_depth=2)
    W = OM.init_hf_weights(cfg)
    batch = _long_batch(cfg, 4 if ragged else 3, seed=11, ragged=ragged)
    l0, g0, a0, _ = _run(dev, cfg, W, batch, prune=False)
    l1, g1, a1, n1 = _run(dev, cfg, W, batch, prune=True)
    # `last_pruned` reports what the stack DID (round-5 advisor): the frozen MoE trunk prunes (top-1 gather / scatter branch), the dense frozen
    # trunk ignores the row set and must say so — for it the comparison below is the identity of two unpruned runs
    assert not a0 and a1 == moe and (n1 > 0) == moe, "the pruned path did not engage where it should (or claims to where it does not)"
    for k in l0:
        assert torch.equal(l0[k], l1[k]), (k, l0[k], l1[k])
    assert g0.keys() == g1.keys() and len(g0) > 0
    for n in g0:
        assert torch.equal(g0[n], g1[n]), n
    print(f"frozen MoE trunk: {n1} read rows; {len(l0)} losses and {len(g0)} gradients bit-identical")


@pytest.mark.parametrize("targets,p,ragged", [("gate_proj,up_proj,down_proj", 0.0, False), ("q_proj,k_proj,v_proj,o_proj,gate_proj,up_proj,down_proj", 0.0, False),
                                              ("gate_proj,up_proj,down_proj", 0.0, True)])      # ragged: right-padded samples (key padding mask on)
def test_lora_last_layer_mlp_on_read_rows(dev, targets, p, ragged):
    cfg = MedPLIBConfig.tiny(moe_enable=False, sam_depth=2, num_hidden_layers=2)
    W = OM.init_hf_weights(cfg)
    batch = _long_batch(cfg, 4 if ragged else 3, seed=12, ragged=ragged)

    def init(lora):
        g = torch.Generator().manual_seed(31)
        return [(torch.randn(p_.shape, generator=g) * (0.05 if "lora_A" in n else 0.03)).to(torch.bfloat16).float() for n, p_ in zip(lora.names, lora.params)]
    kw = dict(lora_r=8, lora_alpha=16, lora_dropout=p, lora_target_modules=targets)
    l0, g0, a0, _ = _run(dev, cfg, W, batch, False, kw, init)
    l1, g1, a1, n1 = _run(dev, cfg, W, batch, True, kw, init)
    assert not a0 and a1 and n1 > 0
    for k in l0:
        np.testing.assert_allclose(l1[k].numpy(), l0[k].numpy(), rtol=2e-6, atol=2e-6, err_msg=k)
    assert g0.keys() == g1.keys() and len(g0) > 0
    worst = 0.0
    for n in g0:
        scale = g0[n].abs().max().item()
        err = (g0[n] - g1[n]).abs().max().item()
        worst = max(worst, err / (scale + 1e-20))
        # the same products summed in fp32 over fewer rows (the dropped rows contributed exact zeros): fp32 summation order only
        assert err <= 1e-5 * scale + 1e-12, (n, err, scale)
    print(f"LoRA {targets}: {n1} read rows; worst gradient difference {worst:.2e} of the gradient's largest entry")


@pytest.mark.parametrize("targets,p", [("gate_proj,up_proj,down_proj", 0.05), ("q_proj,k_proj,v_proj,o_proj,gate_proj,up_proj,down_proj", 0.0)])
def test_lora_weight_gradients_on_a_side_stream_are_bit_identical(dev, targets, p):
    """MP_LORA_WGRAD_STREAM=1 (llama_lora.backward): the adapters' dA^T products and the unpack into the flat gradient buffer run on their own
    stream — the same kernels on the same operands in the same order per gradient, so every loss and every gradient is the same bit for bit (two
    independent runs per setting: the comparison also holds run against run)."""
    from medplib_amd.model import llama_lora
    cfg = MedPLIBConfig.tiny(moe_enable=False, sam_depth=2, num_hidden_layers=3)
    W = OM.init_hf_weights(cfg)
    batch = _long_batch(cfg, 3, seed=13)

    def init(lora):
        g = torch.Generator().manual_seed(32)
        return [(torch.randn(p_.shape, generator=g) * (0.05 if "lora_A" in n else 0.03)).to(torch.bfloat16).float() for n, p_ in zip(lora.names, lora.params)]
    kw = dict(lora_r=8, lora_alpha=16, lora_dropout=p, lora_target_modules=targets)
    res = {}
    for side in (False, True):
        llama_lora._WGRAD_STREAM = side
        try:
            res[side] = [_run(dev, cfg, W, batch, True, kw, init) for _ in range(2)]
        finally:
            llama_lora._WGRAD_STREAM = False
    for (l0, g0, _, _), (l1, g1, _, _) in zip(res[False], res[True]):
        for k in l0:
            assert torch.equal(l0[k], l1[k]), k
        assert g0.keys() == g1.keys() and len(g0) > 0
        for n in g0:
            assert torch.equal(g0[n], g1[n]), n
