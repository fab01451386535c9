"""Checkpoint converters around the hot path — host-side state-dict plumbing, no kernels.

* `merge_deepspeed_states(directory)`: what the reference's `params_bf16_to_f32.py:5-28` does with a DeepSpeed checkpoint
  directory — every `*model_states.pt` file is read; the dense file keeps its parameters under `['module']`, the per-expert files
  (`layer_{L}_expert_{e}_mp_rank_00_model_states.pt`, "expert" in the name) ARE the state dict; all are merged into one HF-layout
  dict in fp32, duplicate keys are an error.  The result loads through `MedPLIBForCausalLM.load_hf_state_dict`.
* `moe_layer_indices(...)` / `seed_experts_from_dense(...)`: `initialize_moe_modules` (medplib_moe_llama.py:572-638) as a
  state-dict transformation: the MoE layers are chosen by `moe_mode` / `moe_layers_idx`, expert e of MoE layer L receives the
  dense MLP of source checkpoint e (`model.layers.{L}.mlp.{gate,up,down}_proj.weight`), the dense MLP keys of those layers
  disappear, and the gate `…deepspeed_moe.gate.wg.weight` (fp32 [E, hidden]) is new (nn.Linear default init, seeded here)."""
import math
import os
from typing import Dict, List, Optional, Sequence

import torch

_PROJ = ("gate_proj", "up_proj", "down_proj")


def merge_deepspeed_states(directory: str, dtype=torch.float32) -> Dict[str, torch.Tensor]:
    combined = {}
    for fn in sorted(os.listdir(directory)):
        if not fn.endswith("model_states.pt"):
            continue
        blob = torch.load(os.path.join(directory, fn), map_location="cpu")
        sd = blob if "expert" in fn else blob["module"]
        for k, v in sd.items():
            if k in combined:
                raise ValueError(f"Duplicate key found in state dicts: {k}")
            combined[k] = v.to(dtype) if torch.is_tensor(v) and v.is_floating_point() else v
    return combined


def moe_layer_indices(num_layers: int, moe_mode: str = "dense", moe_layers_idx: Optional[Sequence[int]] = None) -> List[int]:
    """medplib_moe_llama.py:575-596."""
    if moe_layers_idx is not None:
        idx = list(moe_layers_idx)
        assert len(idx) <= num_layers and max(idx) < num_layers and min(idx) >= 0
        return idx
    if moe_mode == "first_half":
        return list(range(0, num_layers // 2))
    if moe_mode == "second_half":
        return list(range(num_layers // 2, num_layers))
    if moe_mode == "sparse":
        return list(range(num_layers))[::2]
    if moe_mode == "dense":
        return list(range(num_layers))
    raise NotImplementedError(f'Only support ["first_half", "second_half", "sparse", "dense"], but found {moe_mode}')


def seed_experts_from_dense(base: Dict[str, torch.Tensor], expert_sources: Sequence[Dict[str, torch.Tensor]], num_experts: Sequence[int],
                            layers: Sequence[int], hidden_size: int, gate_seed: int = 0) -> Dict[str, torch.Tensor]:
    """`base`: the dense HF-layout checkpoint the MoE model starts from; `expert_sources[e]`: the (dense) checkpoint whose MLPs seed
    expert e (the reference passes one state dict per expert: `expert_state_dict[e_idx]`, :629-637); `num_experts`: one entry per MoE
    layer (a single entry is broadcast, :597-598)."""
    layers = list(layers)
    ne = list(num_experts) * len(layers) if len(num_experts) == 1 else list(num_experts)
    assert len(ne) == len(layers)
    out = dict(base)
    g = torch.Generator().manual_seed(gate_seed)
    for E, L in zip(ne, layers):
        assert len(expert_sources) >= E, f"layer {L}: {E} experts need {E} source checkpoints, got {len(expert_sources)}"
        for p in _PROJ:
            out.pop(f"model.layers.{L}.mlp.{p}.weight", None)
        for e in range(E):
            for p in _PROJ:
                out[f"model.layers.{L}.mlp.deepspeed_moe.experts.deepspeed_experts.{e}.{p}.weight"] = \
                    expert_sources[e][f"model.layers.{L}.mlp.{p}.weight"].clone()
        bound = 1.0 / math.sqrt(hidden_size)           # nn.Linear(hidden, E, bias=False) default init, kept in fp32 (TopKGate.wg)
        out[f"model.layers.{L}.mlp.deepspeed_moe.gate.wg.weight"] = (torch.rand(E, hidden_size, generator=g) * 2 - 1) * bound
    return out
