"""The reference's MODULE SURFACE over the HIP path (Face 1): what `train_ds_medplib.py` and `vqa_infer.py` call on
`model.MedPLIB.MedPLIBForCausalLM` / `model.LISA.LISAForCausalLM`, so the drivers' call sequence runs unchanged:

    model = Cls.from_pretrained(path, torch_dtype=..., low_cpu_mem_usage=True, ignore_mismatched_sizes=True, **vars(args))   # :225-232
    model.config.eos_token_id = ...; model.enable_input_require_grads(); model.gradient_checkpointing_enable()             # :233-238
    model.get_model().initialize_vision_modules(cfg); model.get_model().initialize_bird_modules(cfg)                       # :241-247
    vision_tower = model.get_model().get_vision_tower(); vision_tower.to(dtype=..., device=...)                            # :250-251
    for p in vision_tower.parameters(): p.requires_grad = False; ... model.get_model().mm_projector.parameters()           # :253-256
    find_linear_layers(model, targets)  # isinstance(module, nn.Linear) over model.named_modules()                         # :265-285
    model = get_peft_model(model, LoraConfig(...))          # medplib_amd.peft_compat                                       # :294-303
    model.initialize_moe_modules(args); model.resize_token_embeddings(len(tokenizer))                                      # :310-312
    for n, p in model.named_parameters(): if any(x in n for x in sft_modules): p.requires_grad = True                      # :316-326
    engine, optimizer, loader, scheduler = deepspeed.initialize(model=model, model_parameters=model.parameters(), ...)     # :439-448

How it maps onto this build.  The frozen trunk lives in fused kernel-layout buffers (qkv fused, gate|up interleaved, experts stacked,
convolutions as im2col matrices), so a reference parameter such as `model.layers.3.mlp.gate_proj.weight` is not a tensor that exists.
`named_parameters()` / `named_modules()` therefore walk a SKELETON: an nn.Module tree with the reference's names, classes (nn.Linear /
nn.Embedding where the reference has them — LoRA-target discovery is an isinstance + substring test) and shapes, whose frozen leaves are
HANDLES (meta-device Parameters: name, shape, dtype, requires_grad — no storage), and whose trainable tail (`text_hidden_fcs`,
`mask_decoder`, `prompt_encoder`) are the REAL modules.  The drivers only flip `requires_grad` and count `numel()` on trunk
parameters; values are read through `state_dict()` (the reference's key layout, materialised from the fused buffers).
`engine.initialize` calls `resolve_training_plan()`, which turns the flags into this build's training state: LoRA adapters and the
`--sft_modules` families (`lm_head`, `embed_tokens`, `input_layernorm`, `post_attention_layernorm`, `wg`, `mm_projector`,
`mm_token_compressor`, `region_fea_adapter`, `mask_encoder`) become real fp32 Parameters (`enable_lora`), which replace their
handles in the skeleton; a trainable flag on anything else (a frozen tower, a base projection) raises — loudly, never ignored."""
import glob
import json
import os
import types

import torch
import torch.nn as nn

from . import checkpoint as CK
from .model import medplib as core
from .model.config import MedPLIBConfig
from .model.llama import LlamaStack

SFT_FAMILIES = ("lm_head", "embed_tokens", "input_layernorm", "post_attention_layernorm", "wg", "mm_projector", "mm_token_compressor",
                "region_fea_adapter", "mask_encoder")
_PROJ_IO = {"q_proj": "dd", "k_proj": "dd", "v_proj": "dd", "o_proj": "dd", "gate_proj": "df", "up_proj": "df", "down_proj": "fd"}


def _handle(shape, dtype, requires_grad=True):
    return nn.Parameter(torch.empty(tuple(shape), dtype=dtype, device="meta"), requires_grad=requires_grad)


class _Node(nn.Module):
    """A skeleton module that only carries names (children / handle parameters)."""


class _LoraLinear(nn.Module):
    """What peft 0.10 turns a targeted nn.Linear into: `.base_layer`, `.lora_A.default`, `.lora_B.default` (+ dropout)."""

    def __init__(self, base, r, dropout):
        super().__init__()
        self.base_layer = base
        self.lora_dropout = nn.ModuleDict({"default": nn.Dropout(dropout)})
        self.lora_A = nn.ModuleDict({"default": nn.Linear(base.in_features, r, bias=False, device="meta", dtype=torch.float32)})
        self.lora_B = nn.ModuleDict({"default": nn.Linear(r, base.out_features, bias=False, device="meta", dtype=torch.float32)})
        base.weight.requires_grad = False


def _tree_from_keys(spec):
    """{dotted name: (shape, dtype)} -> nested _Node tree with handle leaves (used for the towers, whose internals the drivers never
    address by class)."""
    root = _Node()
    for name, (shape, dtype) in spec.items():
        parts = name.split(".")
        node = root
        for p in parts[:-1]:
            if p not in node._modules:
                node.add_module(p, _Node())
            node = node._modules[p]
        node.register_parameter(parts[-1], _handle(shape, dtype))
    return root


class VisionTowerSurface:
    """`model.get_model().get_vision_tower()` (clip_encoder.py:6-100): `.image_processor`, `.to()`, `.parameters()`, `.is_loaded`,
    `.hidden_size`, `.num_patches`, `.config`, callable on images."""

    def __init__(self, owner):
        self._owner = owner
        self.is_loaded = True
        self._processor = None

    @property
    def image_processor(self):
        if self._processor is None:
            from transformers import CLIPImageProcessor         # pure preprocessing object; no weights, no network
            s = self._owner.config.clip_image_size
            from .preprocess import CLIP_MEAN, CLIP_STD
            self._processor = CLIPImageProcessor(size={"shortest_edge": s}, crop_size={"height": s, "width": s}, do_center_crop=True,
                                                 do_normalize=True, do_resize=True, image_mean=list(CLIP_MEAN), image_std=list(CLIP_STD),
                                                 resample=3)
        return self._processor

    def to(self, *args, **kwargs):
        return self                                          # the tower lives on the model's device in bf16 already

    def parameters(self):
        return (p for n, p in self._owner.named_parameters() if ".vision_tower." in "." + n)

    def requires_grad_(self, flag=True):
        for p in self.parameters():
            p.requires_grad = flag
        return self

    @property
    def hidden_size(self):
        return self._owner.config.clip_hidden_size

    @property
    def num_patches(self):
        return self._owner.config.clip_num_patches

    @property
    def config(self):
        c = self._owner.config
        return types.SimpleNamespace(hidden_size=c.clip_hidden_size, image_size=c.clip_image_size, patch_size=c.clip_patch_size,
                                     num_hidden_layers=c.clip_num_layers, num_attention_heads=c.clip_num_heads)

    def __call__(self, images):
        """CLIPVisionTower.forward: [n,3,H,W] -> [n, patches, C] features of hidden_states[select_layer] without CLS."""
        c = self._owner.config
        _, raw = self._owner.model.vision_tower.encode_images(images, return_raw=True)
        return raw.view(images.shape[0], c.clip_num_patches, c.clip_hidden_size)


class _ProjectorSurface:
    def __init__(self, owner):
        self._owner = owner

    def parameters(self):
        return (p for n, p in self._owner.named_parameters() if ".mm_projector." in "." + n)


class InnerSurface:
    """`model.get_model()` (MedPLIBModel / LisaModel, medplib_arch.py:111-195, MedPLIB.py:120-190)."""

    def __init__(self, owner):
        self._owner = owner

    @property
    def config(self):
        return self._owner.config

    def __getattr__(self, name):                              # text_hidden_fcs, visual_model, llm, ... of the real inner module
        return getattr(self._owner.model, name)

    def get_vision_tower(self):
        return self._owner._tower_surface

    @property
    def mm_projector(self):
        return _ProjectorSurface(self._owner)

    def initialize_vision_modules(self, model_args, fsdp=None):
        """medplib_arch.py:147-186: (re)load the CLIP tower named by `model_args.vision_tower` and an optional
        `pretrain_mm_mlp_adapter` file.  A directory that is not on disk (a hub id) keeps what from_pretrained() loaded."""
        own = self._owner
        path = getattr(model_args, "vision_tower", None) or getattr(model_args, "mm_vision_tower", None)
        if isinstance(path, str) and os.path.isdir(path):
            sd = _read_weight_files(path)
            if sd:
                pre = "model.vision_tower.vision_tower."
                cur = own._current_hf_state()
                for k, v in sd.items():
                    k2 = pre + (k if k.startswith("vision_model.") else "vision_model." + k)
                    if k2 in cur and tuple(cur[k2].shape) == tuple(v.shape):
                        cur[k2] = v
                own.model.vision_tower.load_hf(cur)
        adapter = getattr(model_args, "pretrain_mm_mlp_adapter", None)
        if adapter:
            w = torch.load(adapter, map_location="cpu")
            cur = own._current_hf_state()
            for k, v in w.items():
                if "mm_projector" in k:
                    cur["model.mm_projector." + k.split("mm_projector.")[1]] = v
            own.model.vision_tower.load_hf(cur)

    def initialize_bird_modules(self, config):
        """MedPLIB.py:141-164 / LISA.py:131-157: SAM-Med2D from `vision_pretrained` (torch.load(path)['model'], non-strict), frozen
        except the mask decoder when `train_mask_decoder`; `text_hidden_fcs` trainable."""
        own = self._owner
        path = getattr(own, "vision_pretrained", None)
        if path and os.path.exists(path):
            own.load_sam_state_dict(torch.load(path, map_location="cpu")["model"])
        for n, p in own.named_parameters():
            if ".visual_model." in "." + n:
                p.requires_grad = bool(getattr(config, "train_mask_decoder", True)) and ".mask_decoder." in n
            elif ".text_hidden_fcs." in "." + n:
                p.requires_grad = True

    initialize_lisa_modules = initialize_bird_modules


def _read_weight_files(path):
    """Every weight file of an HF-style directory merged into one dict: pytorch_model*.bin (single or sharded), *.safetensors."""
    sd = {}
    for f in sorted(glob.glob(os.path.join(path, "*.bin"))):
        if os.path.basename(f).startswith("training_args"):
            continue
        sd.update(torch.load(f, map_location="cpu"))
    for f in sorted(glob.glob(os.path.join(path, "*.safetensors"))):
        from safetensors.torch import load_file
        sd.update(load_file(f))
    return sd


def config_from_pretrained(path, kwargs):
    """config.json of an HF LLaVA / MedPLIB directory (or one written by this build's save_pretrained) -> MedPLIBConfig.  The CLIP
    dims come from the `mm_vision_tower` directory's own config.json when it is on disk (the LLaVA config only names the tower)."""
    raw = json.load(open(os.path.join(path, "config.json")))
    fields = set(MedPLIBConfig.__dataclass_fields__)
    kw = {k: v for k, v in raw.items() if k in fields and v is not None}
    for k in ("sam_global_attn",):
        if k in kw:
            kw[k] = tuple(kw[k])
    if "rope_theta" not in kw and isinstance(raw.get("rope_parameters"), dict):
        kw["rope_theta"] = raw["rope_parameters"].get("rope_theta", 10000.0)
    tower = kwargs.get("vision_tower") or raw.get("mm_vision_tower") or raw.get("vision_tower")
    if isinstance(tower, str) and os.path.exists(os.path.join(tower, "config.json")):
        t = json.load(open(os.path.join(tower, "config.json")))
        t = t.get("vision_config", t)
        for src, dst in (("image_size", "clip_image_size"), ("patch_size", "clip_patch_size"), ("hidden_size", "clip_hidden_size"),
                         ("intermediate_size", "clip_intermediate_size"), ("num_hidden_layers", "clip_num_layers"),
                         ("num_attention_heads", "clip_num_heads"), ("layer_norm_eps", "clip_ln_eps")):
            if src in t:
                kw[dst] = t[src]
    moe = raw.get("moe")
    if isinstance(moe, dict) and moe.get("moe_enable"):        # a trained MoE checkpoint (medplib_moe_llama.py:63-78)
        kw.update(moe_enable=True, moe_layers_idx=moe.get("moe_layers_idx"), top_k_experts=moe.get("top_k_experts", 1),
                  capacity_factor=moe.get("capacity_factor", 1.5), eval_capacity_factor=moe.get("eval_capacity_factor", 2.0),
                  min_capacity=moe.get("min_capacity", 0), use_residual=moe.get("use_residual", False),
                  router_aux_loss_coef=moe.get("router_aux_loss_coef", 0.0))
        ne = moe.get("num_experts")
        if ne:
            kw["num_experts"] = int(ne[0] if isinstance(ne, (list, tuple)) else ne)
    elif "moe_enable" not in raw:
        kw["moe_enable"] = False                               # a dense LLaVA checkpoint: initialize_moe_modules() converts it later
    return MedPLIBConfig(**kw)


class SurfaceMixin:
    """Mixed in front of the core classes by model/MedPLIB.py and model/LISA.py."""

    # ------------------------------------------------------------------ construction
    @classmethod
    def from_pretrained(cls, pretrained_model_name_or_path, *model_args, torch_dtype=None, low_cpu_mem_usage=True,
                        ignore_mismatched_sizes=False, device=None, **kwargs):
        """`Cls.from_pretrained(args.version, torch_dtype=, low_cpu_mem_usage=True, ignore_mismatched_sizes=True, **vars(args))`
        (train_ds_medplib.py:225-232; vqa_infer.py:244-256).  Keys the directory does not hold (a LLaVA base has no SAM /
        text_hidden_fcs) keep their initialisation, like HF's non-strict load; a size mismatch is an error unless
        `ignore_mismatched_sizes` (then that tensor keeps its initialisation, as in HF)."""
        path = pretrained_model_name_or_path
        if not os.path.isdir(path):
            raise FileNotFoundError(f"{path}: from_pretrained needs a local HF-layout directory (no network on this path)")
        if torch_dtype not in (None, torch.bfloat16):
            raise ValueError("this build computes the trunk in bf16 (the reference's --precision bf16); torch_dtype must be torch.bfloat16")
        cfg = config_from_pretrained(path, kwargs)
        if device is None:
            lr = kwargs.get("local_rank")
            device = torch.device("cuda", int(lr) if lr is not None else torch.cuda.current_device())
        model = cls(cfg, device=device, **kwargs)
        model.vision_pretrained = kwargs.get("vision_pretrained")
        sd = _read_weight_files(path)
        model.load_state_dict(sd, strict=False, ignore_mismatched_sizes=ignore_mismatched_sizes)
        return model

    def _surface_init(self):
        self._sk = {"tree": None}              # a dict, so nn.Module.__setattr__ does not register the skeleton as a child
        self._flags = {}                       # reference name -> requires_grad, survives skeleton rebuilds
        self._lora_cfg = None                  # (r, alpha, dropout, targets) once get_peft_model() wrapped the model
        self._tower_surface = VisionTowerSurface(self)
        self._resolved = False
        self.vision_pretrained = None

    def to(self, *args, **kwargs):
        """`model.to(dtype=torch_dtype, device=args.local_rank)` of the inference driver (vqa_infer.py:237-238).  The model was built on
        its GPU with the kernel-layout dtypes (bf16 trunk, fp32 norm weights / gate / trainable tail); a blanket cast would break
        those layouts, so this validates the request and changes nothing: bf16 (or no dtype) on the device the model lives on."""
        device, dtype, _, _ = torch._C._nn._parse_to(*args, **kwargs)
        if dtype not in (None, torch.bfloat16):
            raise ValueError(f"this build computes the trunk in bf16 (the reference's --precision bf16); .to(dtype={dtype}) is not available")
        if device is not None:
            want = torch.device(device)
            if want.type != "cuda" or (want.index is not None and want.index != self.device_.index):
                raise NotImplementedError(f"the model lives on {self.device_}; build it there (from_pretrained(..., local_rank=) / device=) "
                                          f"instead of moving it to {want}")
        return self

    def get_model(self):
        return InnerSurface(self)

    def get_vision_tower(self):
        return self._tower_surface

    def enable_input_require_grads(self):
        """HF hook that makes the embedding output require grad so checkpointed blocks get gradients (train_ds_medplib.py:237).
        Nothing to do: this build's decoder backward (LlamaLoRAFn) is explicit and starts from the spliced embeddings."""

    def gradient_checkpointing_enable(self, *a, **k):
        """train_ds_medplib.py:238.  Accepted: the forward keeps the ~0.7 GB per layer the backward reads (22 GB of 288 at 7B, batch 8),
        so there is no recomputation to enable — the reference recomputes to fit 40/80 GB cards."""

    # ------------------------------------------------------------------ state dicts in the reference's key layout
    def _current_hf_state(self):
        lora, self.model.lora = getattr(self.model, "lora", None), None     # hf_state_dict() refuses unmerged adapters: export the base
        try:
            return dict(core.MedPLIBForCausalLM.hf_state_dict(self))
        finally:
            self.model.lora = lora

    def state_dict(self, *a, **k):
        return self._current_hf_state()

    def load_state_dict(self, sd, strict=True, ignore_mismatched_sizes=False):
        cur = self._current_hf_state()
        missing = [k for k in cur if k not in sd]
        unexpected, mismatched = [], []
        for k, v in sd.items():
            k2 = k if k in cur else k.replace("vision_tower.vision_tower.", "vision_tower.vision_tower.vision_model.")   # 5.x CLIP names
            if k2 not in cur:
                unexpected.append(k)
            elif tuple(cur[k2].shape) != tuple(v.shape):
                mismatched.append(k)
                if not ignore_mismatched_sizes:
                    raise RuntimeError(f"size mismatch for {k}: checkpoint {tuple(v.shape)} vs model {tuple(cur[k2].shape)}")
            else:
                cur[k2] = v
        if strict and (missing or unexpected):
            raise RuntimeError(f"load_state_dict: missing {missing[:5]}..., unexpected {unexpected[:5]}...")
        self.load_hf_state_dict(cur)
        return types.SimpleNamespace(missing_keys=missing, unexpected_keys=unexpected, mismatched_keys=mismatched)

    # ------------------------------------------------------------------ MoE conversion, vocabulary
    def initialize_moe_modules(self, model_args):
        """medplib_moe_llama.py:488-649 on the fused buffers: the MLP of every chosen layer becomes E experts (seeded from the dense
        MLPs of the `expert_pretrained_path` checkpoints — expert e from directory e — or, without them, E copies of the layer's own
        MLP, which is what DeepSpeed's MoE(expert=mlp) deep copy gives), plus a fresh fp32 gate `wg`.  `text_hidden_fcs`, the mask
        decoder (first directory) and `region_fea_adapter` (later directories) are taken over as the reference does (:530-560).
        `expert_pretrained_path=None` is accepted (the reference's ICL script crashes on it)."""
        cfg = self.config
        ne = getattr(model_args, "num_experts", [cfg.num_experts])
        ne = list(ne) if isinstance(ne, (list, tuple)) else [int(ne)]
        if len(set(ne)) != 1:
            raise NotImplementedError("a different expert count per MoE layer is not built (the shipped scripts use one value)")
        E = int(ne[0])
        layers = CK.moe_layer_indices(cfg.num_hidden_layers, getattr(model_args, "moe_mode", "dense"), getattr(model_args, "moe_layers_idx", None))
        base = self._current_hf_state()
        paths = getattr(model_args, "expert_pretrained_path", None)
        sources = []
        for idx, p in enumerate(paths.split(",") if paths else []):
            assert os.path.exists(p), f"{p} does not exist"
            src = _read_weight_files(p)
            sources.append(src)
            take = ("text_hidden_fcs", "mask_decoder") if idx == 0 else ("region_fea_adapter",)
            for k, v in src.items():
                if any(t in k for t in take) and k in base and tuple(base[k].shape) == tuple(v.shape):
                    base[k] = v
        if sources and len(sources) < E:
            raise ValueError(f"{E} experts need {E} expert_pretrained_path directories, got {len(sources)}")
        if not sources:
            sources = [base] * E
        seeded = CK.seed_experts_from_dense(base, sources[:E], [E], layers, cfg.hidden_size, gate_seed=int(getattr(model_args, "seed", 0) or 0))
        cfg.moe_enable, cfg.num_experts, cfg.moe_layers_idx = True, E, list(layers)
        for k in ("top_k_experts", "capacity_factor", "eval_capacity_factor", "min_capacity", "use_residual", "router_aux_loss_coef"):
            if getattr(model_args, k, None) is not None:
                setattr(cfg, k, getattr(model_args, k))
        if cfg.use_residual:
            raise NotImplementedError("initialize_moe_modules(use_residual=True): seed the residual MLP through load_state_dict instead")
        self._remember_flags()
        inherit = {}                                          # MoE(expert=mlp) deep-copies the (maybe LoRA-wrapped) MLP: flags travel
        for L in layers:
            for name, flag in self._flags.items():
                pre = f"model.layers.{L}.mlp."
                if name.startswith(pre) and "deepspeed_moe" not in name:
                    for e in range(E):
                        inherit[f"{pre}deepspeed_moe.experts.deepspeed_experts.{e}.{name[len(pre):]}"] = flag
        self._flags.update(inherit)
        self.model.llm = LlamaStack(cfg, self.device_)
        self.model.llm.training = self.training
        self.load_hf_state_dict(seeded)
        self._sk["tree"] = None
        ep = int(getattr(model_args, "ep_size", 1) or 1)
        if ep > 1:
            from .expert_parallel import ExpertParallel, build_groups, build_host_group
            group, _ = build_groups(ep)
            self.model.llm.enable_expert_parallel(ExpertParallel(group, ep, E, host_group=build_host_group(ep)))

    def resize_token_embeddings(self, new_num_tokens=None, **_):
        """HF resize (train_ds_medplib.py:312): the first min(old, new) rows are kept, new rows are drawn N(0, 0.02) like
        `_init_weights` does for the grown nn.Embedding / lm_head of transformers 4.31."""
        llm = self.model.llm
        old = llm.embed_tokens.shape[0]
        if new_num_tokens is None or new_num_tokens == old:
            return self
        g = torch.Generator(device=self.device_).manual_seed(4242)
        for name in ("embed_tokens", "lm_head"):
            t = getattr(llm, name)
            new = (torch.randn(new_num_tokens, t.shape[1], generator=g, device=self.device_, dtype=torch.float32) * 0.02).to(t.dtype)
            k = min(old, new_num_tokens)
            new[:k] = t[:k]
            setattr(llm, name, new)
        self.config.vocab_size = int(new_num_tokens)
        self._invalidate()
        return self

    # ------------------------------------------------------------------ the skeleton
    def _build_skeleton(self):
        cfg, m = self.config, self.model
        d, ff, V = cfg.hidden_size, cfg.intermediate_size, m.llm.embed_tokens.shape[0]
        bf, f32 = torch.bfloat16, torch.float32
        lora = self._lora_cfg
        real = {}
        if getattr(m, "lora", None) is not None:              # resolved: adapters / sft families are real fp32 Parameters now
            real = {n: p for n, p in zip(m.lora.names, m.lora.params)}

        def linear(i, o, name, dtype=bf, bias=False):
            lin = nn.Linear(i, o, bias=bias, device="meta", dtype=dtype)
            short = name.rsplit(".", 1)[-1]
            if lora is not None and short in lora["targets"] and not any(x in name for x in ("visual_model", "vision_tower", "mm_projector")):
                return _LoraLinear(lin, lora["r"], lora["dropout"])
            return lin

        def mlp(prefix):
            n = _Node()
            for t in ("gate_proj", "up_proj", "down_proj"):
                i, o = (d, ff) if _PROJ_IO[t] == "df" else (ff, d)
                n.add_module(t, linear(i, o, prefix + t))
            return n

        def norm():
            n = _Node()
            n.register_parameter("weight", _handle((d,), bf))
            return n
        root, inner = _Node(), _Node()
        root.add_module("model", inner)
        emb = nn.Embedding(V, d, device="meta", dtype=bf)
        inner.add_module("embed_tokens", emb)
        layers = nn.ModuleList()
        for i in range(cfg.num_hidden_layers):
            lay, att = _Node(), _Node()
            p = f"model.layers.{i}."
            for t in ("q_proj", "k_proj", "v_proj", "o_proj"):
                att.add_module(t, linear(d, d, p + "self_attn." + t))
            lay.add_module("self_attn", att)
            if i in m.llm.moe_layers:
                moe, dsm, gate, experts = _Node(), _Node(), _Node(), _Node()
                gate.add_module("wg", nn.Linear(d, cfg.num_experts, bias=False, device="meta", dtype=f32))
                experts.add_module("deepspeed_experts", nn.ModuleList(
                    mlp(p + f"mlp.deepspeed_moe.experts.deepspeed_experts.{e}.") for e in range(cfg.num_experts)))
                dsm.add_module("gate", gate); dsm.add_module("experts", experts)
                moe.add_module("deepspeed_moe", dsm)
                if cfg.use_residual:
                    moe.add_module("mlp", mlp(p + "mlp.mlp."))
                    moe.add_module("coefficient", nn.Linear(d, 2, device="meta", dtype=bf))
                lay.add_module("mlp", moe)
            else:
                lay.add_module("mlp", mlp(p + "mlp."))
            lay.add_module("input_layernorm", norm()); lay.add_module("post_attention_layernorm", norm())
            layers.append(lay)
        inner.add_module("layers", layers)
        inner.add_module("norm", norm())
        # towers and ICL modules: names / shapes from their own exporters (small), as plain name trees
        small = dict(m.vision_tower.export_hf())
        small.update(m.visual_model.image_encoder.export_ref("model.visual_model.image_encoder."))
        if m.mm_token_compressor is not None:
            small.update(m.mm_token_compressor.export_hf())
        if m.mask_encoder is not None:
            small.update(m.mask_encoder.export_hf())
        spec = {k[len("model."):]: (tuple(v.shape), v.dtype) for k, v in small.items()}
        side = _tree_from_keys(spec)
        for name, child in side._modules.items():
            if name == "visual_model":
                vm = child
                vm.add_module("prompt_encoder", m.visual_model.prompt_encoder)       # the REAL trainable-tail modules
                vm.add_module("mask_decoder", m.visual_model.mask_decoder)
            inner.add_module(name, child)
        # region_fea_adapter / mm_projector are nn.Linear / Sequential(Linear, GELU, Linear) in the reference
        inner.add_module("text_hidden_fcs", m.text_hidden_fcs)
        root.add_module("lm_head", linear(d, V, "lm_head"))
        # the peft wrapper does not touch lm_head unless it is a target; restore flags, then splice in real parameters
        for name, p in list(root.named_parameters()):
            if name in real:
                mod = root.get_submodule(name.rsplit(".", 1)[0])
                mod._parameters[name.rsplit(".", 1)[1]] = real[name]
            elif name in self._flags and p.is_meta:
                p.requires_grad = self._flags[name]
        return root

    def _skel(self):
        if self._sk["tree"] is None:
            self._sk["tree"] = self._build_skeleton()
        return self._sk["tree"]

    def _remember_flags(self):
        if self._sk["tree"] is not None:
            for n, p in self._sk["tree"].named_parameters():
                self._flags[n] = p.requires_grad

    def _invalidate(self):
        self._remember_flags()
        self._sk["tree"] = None

    def named_parameters(self, prefix="", recurse=True, remove_duplicate=True):
        for n, p in self._skel().named_parameters(prefix=prefix, recurse=recurse):
            yield n, p

    def parameters(self, recurse=True):
        for _, p in self.named_parameters():
            yield p

    def named_modules(self, memo=None, prefix="", remove_duplicate=True):
        yield prefix, self
        for n, mod in self._skel().named_modules(prefix=prefix):
            if n != prefix:
                yield n, mod

    def modules(self):
        for _, mod in self.named_modules():
            yield mod

    def core_named_parameters(self):
        """This build's own registration names (nn.Module.named_parameters of the core class)."""
        return nn.Module.named_parameters(self)

    # ------------------------------------------------------------------ flags -> training state
    def resolve_training_plan(self, model_parameters=None):
        """Called by engine.initialize: read the requires_grad flags the driver set, build the training state, return the real
        trainable Parameters (DeepSpeed filters `model_parameters` by requires_grad the same way).  `model_parameters` (an iterable
        of parameters or of param-group dicts, train_ds_medplib.py:422-436) only confirms the flags; the model's own table rules."""
        if self._resolved:
            return self._trainable_real()
        self._remember_flags()
        on = [n for n, p in self.named_parameters() if p.requires_grad]
        fam_on, lora_on, tail_on, other = {}, [], [], []
        for n in on:
            if ".lora_A." in n or ".lora_B." in n:
                lora_on.append(n)
            elif "text_hidden_fcs" in n or ".visual_model.mask_decoder." in n:
                tail_on.append(n)
            else:
                fam = next((f for f in SFT_FAMILIES if ("." + f + ".") in "." + n), None)
                if fam is None or ".visual_model." in n or ".vision_tower." in n:
                    other.append(n)
                else:
                    fam_on.setdefault(fam, []).append(n)
        if other:
            raise NotImplementedError("requires_grad=True on parameters this build keeps frozen (full fine-tuning of the decoder's base "
                                      f"projections / the towers is not built): {other[:6]}{' ...' if len(other) > 6 else ''}")
        all_names = [n for n, _ in self.named_parameters()]
        for fam, names in fam_on.items():                     # the reference's substring match always selects a whole family
            members = [n for n in all_names if (("." + fam + ".") in "." + n) and ".lora_" not in n
                       and ".visual_model." not in n and ".vision_tower." not in n]
            if sorted(members) != sorted(names):
                raise NotImplementedError(f"--sft_modules family `{fam}` is trainable only in part ({len(names)} of {len(members)} tensors)")
        if lora_on and self._lora_cfg is None:
            raise RuntimeError("lora parameters without get_peft_model()")
        need_llm = bool(lora_on) or any(f in fam_on for f in SFT_FAMILIES)
        if need_llm:
            lc = self._lora_cfg or dict(r=8, alpha=16, dropout=0.0, targets=())
            if lc["targets"] and not lora_on:
                raise NotImplementedError("adapters attached but frozen")
            self.enable_lora(lc["r"], lc["alpha"], lc["dropout"], tuple(lc["targets"]), train_gate=("wg" in fam_on),
                             sft_modules=tuple(f for f in SFT_FAMILIES if f in fam_on))
        for n, p in self.core_named_parameters():              # the tail is registered under the reference's own names
            if "text_hidden_fcs" in n or ".mask_decoder." in n:
                p.requires_grad = n in tail_on
        self.config.train_mask_decoder = any(".mask_decoder." in n for n in tail_on)
        self._resolved = True
        self._sk["tree"] = None
        return self._trainable_real()

    def _trainable_real(self):
        ps = [p for p in self.model.text_hidden_fcs.parameters() if p.requires_grad]
        ps += [p for p in self.model.visual_model.mask_decoder.parameters() if p.requires_grad]
        if getattr(self.model, "lora", None) is not None:
            ps += list(self.model.lora.parameters())
        return ps

    def print_trainable_parameters(self):
        tr = sum(p.numel() for p in self.parameters() if p.requires_grad)
        al = sum(p.numel() for p in self.parameters())
        print(f"trainable params: {tr:,d} || all params: {al:,d} || trainable%: {100 * tr / max(al, 1):.4f}")
