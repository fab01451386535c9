"""LoRA checkpoints (peft==0.10.0, requirements.txt:80): what `merge_lora_weights_and_save_hf_model_moe.py:270-345` does with
`get_peft_model(...)` + `load_state_dict` + `merge_and_unload()`, as a pure state-dict transformation, so LoRA-trained stage
II-IV checkpoints load into this build's inference path (`load_hf_state_dict`).  Checkpoint plumbing on the host — no kernels.

peft key layout of a wrapped Linear `…<name>`:  `base_model.model.…<name>.base_layer.weight` (the frozen W; older peft:
`…<name>.weight`), `…<name>.lora_A.default.weight` [r, in], `…<name>.lora_B.default.weight` [out, r]; merge:
W += (lora_alpha / r) * B @ A  (peft LoraLayer.get_delta_weight, bias="none").  Training the adapters (the whole decoder backward) lives in
`medplib_amd/model/llama_lora.py`; this module is the checkpoint side."""
import re
from typing import Dict

import torch

_PREFIX = "base_model.model."


def find_lora_targets(keys, lora_target_modules):
    """The reference's `find_linear_layers` filter (train_ds_medplib.py:265-285): Linear weights whose name contains one of
    lora_target_modules and none of visual_model / vision_tower / mm_projector / text_hidden_fcs."""
    out = []
    for k in keys:
        if not k.endswith(".weight"):
            continue
        name = k[:-len(".weight")]
        if any(x in name for x in ("visual_model", "vision_tower", "mm_projector", "text_hidden_fcs")):
            continue
        if any(x in name for x in lora_target_modules):
            out.append(name)
    return sorted(out)


def merge_lora_state_dict(sd: Dict[str, torch.Tensor], lora_alpha: float, lora_r: int = 0, adapter: str = "default") -> Dict[str, torch.Tensor]:
    """peft-wrapped state dict -> plain HF-layout state dict with every adapter folded into its base weight (merge_and_unload).
    `lora_r` = 0 reads the rank from each lora_A (they may differ per module only if the config used rank patterns)."""
    out = {}
    a_keys = [k for k in sd if k.endswith(f".lora_A.{adapter}.weight")]
    for k, v in sd.items():
        if ".lora_A." in k or ".lora_B." in k or ".lora_embedding_" in k:
            continue
        name = k[len(_PREFIX):] if k.startswith(_PREFIX) else k
        out[name.replace(".base_layer.", ".")] = v
    for ka in a_keys:
        mod = ka[:-len(f".lora_A.{adapter}.weight")]
        A, B = sd[ka], sd[mod + f".lora_B.{adapter}.weight"]
        r = lora_r or A.shape[0]
        assert A.shape[0] == r and B.shape[1] == r, (ka, A.shape, B.shape, r)
        name = (mod[len(_PREFIX):] if mod.startswith(_PREFIX) else mod) + ".weight"
        assert name in out, f"no base weight for adapter {mod}"
        W = out[name]
        out[name] = (W.float() + (float(lora_alpha) / r) * (B.float() @ A.float())).to(W.dtype)
    return out
