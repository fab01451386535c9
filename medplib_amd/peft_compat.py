"""`from peft import LoraConfig, get_peft_model` for the reference driver (train_ds_medplib.py:16,286-303) over this build's adapters.

peft 0.10 is not in the image and its arithmetic is not on the HIP path: the adapters are `medplib_amd/model/llama_lora.py`.  This
module is the calling surface only — `LoraConfig` with the fields the driver sets, and `get_peft_model(model, config)`, which does
what peft does to the names and flags the driver looks at afterwards: every base parameter is frozen, each targeted nn.Linear `X`
becomes `X.base_layer` + `X.lora_A.default` + `X.lora_B.default` (trainable), and the parameter names gain the `base_model.model.`
prefix.  The adapter tensors themselves are created when `engine.initialize` resolves the training plan (the MoE conversion that
deep-copies the wrapped MLPs into experts runs after `get_peft_model` in the driver, train_ds_medplib.py:303-310)."""
from dataclasses import dataclass, field
from typing import List, Optional, Union

from .model.llama_lora import ALL_TARGETS

PREFIX = "base_model.model."


@dataclass
class LoraConfig:
    r: int = 8
    lora_alpha: int = 8
    target_modules: Optional[Union[List[str], str]] = None
    lora_dropout: float = 0.0
    bias: str = "none"
    task_type: Optional[str] = None
    modules_to_save: Optional[List[str]] = None
    fan_in_fan_out: bool = False
    init_lora_weights: bool = True


class PeftModel:
    """`PeftModelForCausalLM` as the driver uses it: call, named_parameters / parameters / named_modules with peft's names,
    `print_trainable_parameters`, and attribute access falling through to the wrapped model."""

    def __init__(self, model, peft_config):
        self.__dict__["base_model"] = model
        self.__dict__["peft_config"] = {"default": peft_config}

    def get_base_model(self):
        return self.__dict__["base_model"]

    def __getattr__(self, name):
        return getattr(self.__dict__["base_model"], name)

    def __setattr__(self, name, value):
        setattr(self.__dict__["base_model"], name, value)

    def __call__(self, *a, **k):
        return self.__dict__["base_model"](*a, **k)

    forward = __call__

    def named_parameters(self, *a, **k):
        for n, p in self.__dict__["base_model"].named_parameters(*a, **k):
            yield PREFIX + n, p

    def parameters(self, *a, **k):
        return self.__dict__["base_model"].parameters(*a, **k)

    def named_modules(self, *a, **k):
        yield "", self
        for n, m in self.__dict__["base_model"].named_modules(*a, **k):
            yield (PREFIX + n).rstrip("."), m

    def train(self, mode=True):
        self.__dict__["base_model"].train(mode)
        return self

    def eval(self):
        return self.train(False)

    def merge_and_unload(self):
        return self.__dict__["base_model"].merge_and_unload()

    def print_trainable_parameters(self):
        self.__dict__["base_model"].print_trainable_parameters()


def get_peft_model(model, peft_config):
    if peft_config.bias != "none":
        raise NotImplementedError("LoraConfig(bias != 'none') is not built (the driver passes 'none')")
    names = peft_config.target_modules
    names = names.split(",") if isinstance(names, str) else list(names or ())
    # the driver passes full module names found by find_linear_layers (…self_attn.q_proj, …mlp.gate_proj): targets are their last part
    targets = sorted({n.rsplit(".", 1)[-1] for n in names})
    bad = [t for t in targets if t not in ALL_TARGETS]
    if bad:
        raise NotImplementedError(f"LoRA on {bad}: adapters are built for the decoder projections {ALL_TARGETS}")
    by_target = {t: sum(1 for n in names if n.rsplit(".", 1)[-1] == t) for t in targets}
    n_layers = model.config.num_hidden_layers
    if any("." in n for n in names) and any(c != n_layers for c in by_target.values()):
        raise NotImplementedError(f"LoRA on a subset of the layers ({by_target}); the adapters cover every decoder layer")
    for _, p in model.named_parameters():
        p.requires_grad = False                                # peft freezes the whole base model
    model._invalidate()
    model._lora_cfg = dict(r=int(peft_config.r), alpha=float(peft_config.lora_alpha), dropout=float(peft_config.lora_dropout),
                           targets=tuple(targets))
    return PeftModel(model, peft_config)
