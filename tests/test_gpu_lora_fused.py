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
