"""Merge a LoRA fine-tune into plain weights and save them in the reference's checkpoint layout -- the job of the reference's
`merge_lora_weights_and_save_hf_model_moe.py` (:270-345: get_peft_model -> load_state_dict(strict=False) -> merge_and_unload ->
save_pretrained).

    python -m medplib_amd.merge --version base.bin --weight <fine-tune> --save_path out_dir [--lora_r 8 --lora_alpha 16 ... model flags]

`--weight` is either
  * a state-dict FILE in peft's key layout (`base_model.model.….lora_A.default.weight`, `….base_layer.weight`), as the reference's
    training + `zero_to_fp32` leaves it: folded on the host by `lora.merge_lora_state_dict` (W += alpha/r * B @ A), keys it does not
    hold are kept from `--version`; or
  * a checkpoint DIRECTORY written by this build's `train.py` (`<log_dir>/ckpt_model`, with `latest`): the model is rebuilt with
    the same `--lora_*` / `--sft_modules` flags, the trained parameters are loaded, `merge_and_unload()` folds the adapters.
Output: `<save_path>/pytorch_model.bin` + `config.json` (`MedPLIBForCausalLM.save_pretrained`)."""
import argparse
import os

import torch

from . import lora as lora_ckpt
from .train import build_model, parse_args as train_args


def main(argv=None):
    own = argparse.ArgumentParser(add_help=False)
    own.add_argument("--weight", required=True)
    own.add_argument("--save_path", required=True)
    o, rest = own.parse_known_args(argv)
    args = train_args(rest)
    device = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")))
    torch.cuda.set_device(device)
    torch.manual_seed(args.seed)
    if os.path.isdir(o.weight):
        cfg, model = build_model(args, device)                       # with the adapters / sft modules the run trained
        tag = open(os.path.join(o.weight, "latest")).read().strip()
        ck = torch.load(os.path.join(o.weight, tag, "mp_rank_00_model_states.pt"), map_location="cpu")
        named = dict(model.named_parameters())
        unknown = [n for n in ck["module"] if n not in named]
        assert not unknown, f"checkpoint parameters the model does not have (different --lora_* / --sft_modules flags?): {unknown[:4]}"
        for n, v in ck["module"].items():
            named[n].data.copy_(v)
        if getattr(model.model, "lora", None) is not None:
            model.model.lora.sync_model(model.model.llm)             # bf16 working copies of the fully fine-tuned matrices
        print(f"loaded {len(ck['module'])} trained tensors from {o.weight}/{tag}")
    else:
        lora_r, args.lora_r = args.lora_r, 0                         # plain model; the adapters are folded on the host
        cfg, model = build_model(args, device)
        sd = torch.load(o.weight, map_location="cpu")
        if any(".lora_A." in k for k in sd):
            sd = lora_ckpt.merge_lora_state_dict(sd, args.lora_alpha, lora_r)
        else:
            sd = {(k[len("base_model.model."):] if k.startswith("base_model.model.") else k): v for k, v in sd.items()}
        full = model.hf_state_dict()
        unexpected = [k for k in sd if k not in full]
        print("unexpected_keys", unexpected)
        print("missing_keys", [k for k in full if k not in sd])      # kept from --version, like load_state_dict(strict=False)
        full.update({k: v for k, v in sd.items() if k in full})
        model.load_hf_state_dict(full)
    model.merge_and_unload()
    model.save_pretrained(o.save_path)
    print(f"saved {o.save_path}/pytorch_model.bin")
    return model


if __name__ == "__main__":
    main()
