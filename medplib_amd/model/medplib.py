"""`MedPLIBForCausalLM` / `LISAForCausalLM` — host-side mirror of model/MedPLIB.py and model/LISA.py over the HIP path.

Same constructor kwargs (model/MedPLIB.py:195-234), same `forward(**batch)` batch-dict contract
(datasets/DataCollatorForSupervisedDataset.py:11-138), same 10-key loss dict (MedPLIB.py:561-572) or
`{pred_masks, gt_masks}` when `inference=True` (MedPLIB.py:507-511), same HF / SAM checkpoint key layout
(SURVEY §8b).  The Python-level serialisation of the reference (per-image SAM loop, per-token seg-mask loop,
per-sample splice loop, per-mask decoder loop) is replaced by batched kernels.

Dead work the reference performs but never consumes is not executed (results are identical — row-wise ops):
full-vocabulary fp32 logits for unsupervised rows, text_hidden_fcs on non-<SEG> rows, the 32 intermediate hidden
states, CLIP's last layer, mask tokens 1-3."""
import math
import os
from typing import List, Optional

import numpy as np
import collections.abc

import torch
import torch.nn as nn

from .. import ops
from . import autograd_ops as A
from .clip import ClipTower
from .config import MedPLIBConfig
from .llama import LlamaStack
from .sam import MaskDecoder, PromptEncoderText, SamImageEncoder
from .icl import MaskTokenEncoder, TokenCompressor
from .splice import IMAGE_TOKEN_INDEX, REGION_TOKEN_INDEX, icl_feature_layout, plan_splice

LOSS_KEYS = ["loss", "ce_loss", "mask_bce_loss", "mask_dice_loss", "mask_loss", "unscale_mask_bce_loss",
             "unscale_mask_dice_loss", "unscale_mask_loss", "unscale_mask_iou_loss", "unscale_mask_focal_loss"]


_PRUNE_LAST_MLP = os.environ.get("MP_PRUNE_LAST_MLP", "1") != "0"
_GATE_MODE = int(os.environ.get("MP_GATE_TOWERS", "0"))                  # 0 off, 1 gate, 2 = record the events only (A/B of the records' own cost)
_GATE_TOWERS = _GATE_MODE >= 1              # A/B: towers that run ahead gate their layers on the previous step's decoder layers (model_forward)


def _h2d(arr, device):
    """Host index array -> device tensor without stalling the host: a pageable-memory copy is synchronous for the caller AND
    ordered behind everything already queued on the stream, so one such copy after the LLM launches makes the host wait for the
    whole stack (measured: 66 of 97 ms per step spent blocked in .to()).  Pinned staging + non_blocking lets the host run ahead and
    queue the SAM encoder / mask tail while the LLM is still executing."""
    t = torch.from_numpy(np.ascontiguousarray(arr))
    return t.pin_memory().to(device, non_blocking=True)


def _np_ids(t):
    if isinstance(t, np.ndarray):
        return t
    return t.detach().cpu().numpy()


class _VisualModel(nn.Module):
    """`model.visual_model` (build_sam_vit_b, model/MedPLIB.py:141-150)."""

    def __init__(self, cfg, device):
        super().__init__()
        self.image_encoder = SamImageEncoder(cfg, device)          # frozen bf16 kernel-layout weights
        self.prompt_encoder = PromptEncoderText(cfg.sam_out_chans, cfg.sam_grid)
        self.mask_decoder = MaskDecoder(cfg.sam_out_chans, cfg.sam_grid)
        self.mask_decoder.fused_bf16_upsampler = bool(cfg.fused_bf16_upsampler)


class _Inner(nn.Module):
    """`model` (MedPLIBModel): LLM stack + vision tower + projector + SAM + text_hidden_fcs."""

    def __init__(self, cfg, device):
        super().__init__()
        self.llm = LlamaStack(cfg, device)
        self.vision_tower = ClipTower(cfg, device)
        self.visual_model = _VisualModel(cfg, device)
        d = cfg.hidden_size
        self.mm_token_compressor = TokenCompressor(d, cfg.mm_compressed_token_count, device) if cfg.mm_token_compress else None
        self.mask_encoder = MaskTokenEncoder(d, cfg.mask_encoder_token_count, device) if cfg.icl_mask_encoder else None
        self.text_hidden_fcs = nn.ModuleList([nn.Sequential(nn.Linear(d, d), nn.ReLU(inplace=True), nn.Linear(d, cfg.out_dim),
                                                            nn.Dropout(0.0))])


class StreamOrderedLosses(collections.abc.Mapping):
    """The loss dict of a step whose mask tail ran on its own stream — a read-only Mapping, NOT a dict subclass: every way of getting
    a value out (`out[k]`, `.get`, `.items()`, `.values()`, `dict(out)`, `{**out}`, `.copy()`, iteration + indexing) goes through
    `__getitem__`, which first makes the READER's current stream wait for the event recorded behind the tail forward (once per
    stream) and tells the caching allocator that this stream uses the tensors (`record_stream`: they were allocated on the tail
    stream).  Any use is therefore ordered exactly as if model_forward had waited itself.  `raw(k)` hands a tensor out without the
    wait, for consumers that run on the tail stream themselves (Engine.backward).  There is no mutation API: dict fast paths that
    bypass `__getitem__` (pop / setdefault / `|`) do not exist on this type."""

    def __init__(self, values, event):
        self._values, self._event, self._ordered = dict(values), event, set()

    def _order(self):
        st = torch.cuda.current_stream()
        if st.cuda_stream not in self._ordered:
            st.wait_event(self._event)
            for v in self._values.values():
                if torch.is_tensor(v) and v.is_cuda:
                    v.record_stream(st)
            self._ordered.add(st.cuda_stream)

    def raw(self, key):
        return self._values[key]

    def __getitem__(self, key):
        if key not in self._values:
            raise KeyError(key)
        self._order()
        return self._values[key]

    def __iter__(self):
        return iter(self._values)

    def __len__(self):
        return len(self._values)

    def copy(self):
        return dict(self)

    def __repr__(self):
        return f"StreamOrderedLosses({list(self._values)})"


class MedPLIBForCausalLM(nn.Module):
    moe_default = True

    def __init__(self, config: Optional[MedPLIBConfig] = None, device="cuda", **kwargs):
        super().__init__()
        cfg = config if config is not None else MedPLIBConfig()
        # kwargs the reference pops / reads (model/MedPLIB.py:195-234); unknown ones are accepted and ignored like there
        for k_cfg, k_kw in (("ce_loss_weight",) * 2, ("dice_loss_weight",) * 2, ("bce_loss_weight",) * 2,
                            ("iou_loss_weight",) * 2, ("focal_loss_weight",) * 2, ("seg_token_idx",) * 2,
                            ("out_dim",) * 2, ("train_mask_decoder",) * 2, ("top_k_experts",) * 2,
                            ("capacity_factor",) * 2, ("eval_capacity_factor",) * 2, ("min_capacity",) * 2,
                            ("router_aux_loss_coef",) * 2, ("moe_layers_idx",) * 2, ("use_residual",) * 2, ("mm_use_im_start_end", "use_mm_start_end"),
                            ("mm_token_compress",) * 2, ("mm_compressed_token_count",) * 2, ("icl_mask_encoder",) * 2,
                            ("mask_encoder_token_count",) * 2):
            if kwargs.get(k_kw) is not None:
                setattr(cfg, k_cfg, kwargs[k_kw])
        if kwargs.get("num_experts") is not None:
            ne = kwargs["num_experts"]
            cfg.num_experts = int(ne[0] if isinstance(ne, (list, tuple)) else ne)
        if not self.moe_default:
            cfg.moe_enable = False
        if cfg.top_k_experts not in (1, 2) and cfg.moe_enable:
            raise ValueError("DeepSpeed MoE supports k = 1 or 2 only (TopKGate asserts the same)")
        self.config = cfg
        self.device_ = torch.device(device)
        self.model = _Inner(cfg, self.device_)
        self.to_device()
        self.seg_token_idx = cfg.seg_token_idx
        self.inference_threshold = 0.1
        self.sam_side_stream = True             # run the SAM encoder beside the LLM stack (model_forward)
        self._sam_stream = None
        # Opt-in (the caller guarantees the image tensors of a step are COMPLETE when it calls model_forward -- resident, or copied on
        # another stream and already synchronised with): the frozen CLIP tower and SAM encoder of the step start on their own streams
        # without queueing behind whatever the calling stream still has in flight, i.e. beside the PREVIOUS step's decoder layers
        # (the host issues a step tens of milliseconds before the GPU reaches it).  The calling stream waits for the CLIP features
        # before the splice, the mask tail for the SAM embedding.  Only while every module in front of the decoder is frozen.
        self.towers_run_ahead = False
        self._vision_stream_obj = None
        # Training only, switched on by engine.initialize(): the fp32 mask tail (forward, and through autograd its backward; the
        # engine adds the optimizer) runs on its own stream, so its ~700 tiny launches overlap the NEXT step's CLIP tower / LLM
        # instead of holding the machine at a few percent occupancy.  The calling stream waits for the tail FORWARD before
        # model_forward returns (the loss dict is safe to read); backward and optimizer stay asynchronous to it and are ordered
        # against the next tail by the tail stream itself.  Anything else that touches trainable state calls sync_side_streams().
        self.decode_with_graph = True            # evaluate(): replay one captured HIP graph per generated token
        self.tail_side_stream = False
        self._tail_stream_obj = None
        self.active_tail_stream = None
        self.capture_intermediates = False      # tests: keep the trunk outputs that feed the trainable tail
        self.last_pruned = None                 # (rows, of rows) the last decoder layer's MLP ran on in the latest training forward, or None
        self.captured = {}

    # ------------------------------------------------------------------ plumbing
    def _side_stream(self):
        if self._sam_stream is None:
            # (A/B, MP_TOWERS_ONE_STREAM=1: the SAM encoder behind the CLIP tower on ONE side stream — one tower kernel in flight at a time)
            name = "clip_tower" if os.environ.get("MP_TOWERS_ONE_STREAM") == "1" else "sam_encoder"
            self._sam_stream = ops.side_stream(self.device_, name, with_gemm_workspace=True)   # own split-K scratch
        return self._sam_stream

    def _vision_stream(self):
        if self._vision_stream_obj is None:
            self._vision_stream_obj = ops.side_stream(self.device_, "clip_tower", with_gemm_workspace=True)
        return self._vision_stream_obj

    def _tail_stream(self):
        if self._tail_stream_obj is None:
            self._tail_stream_obj = ops.side_stream(self.device_, "mask_tail")      # fp32 tail: no bf16 GEMMs, no scratch needed
        return self._tail_stream_obj

    def sync_side_streams(self):
        """Order the calling stream behind everything the side streams have been given (tail backward / optimizer, SAM encoder)."""
        if self._tail_stream_obj is None and self._sam_stream is None and self._vision_stream_obj is None:
            return                                            # nothing was ever put on a side stream (also: host-only surface tests)
        cur = torch.cuda.current_stream()
        for st in (self._tail_stream_obj, self._sam_stream, self._vision_stream_obj):
            if st is not None:
                cur.wait_stream(st)

    def to_device(self):
        nn.Module.to(self, self.device_)
        return self

    def get_model(self):
        return self.model

    TAIL_FAMILIES = ("text_hidden_fcs", "mask_decoder")          # what trains without the decoder backward (the fp32 mask tail)
    DECODER_SIDE_FAMILIES = ("lm_head", "embed_tokens", "input_layernorm", "post_attention_layernorm", "wg", "mm_projector",
                             "mm_token_compressor", "region_fea_adapter", "mask_encoder")

    def trainable_parameters(self, sft_modules=None):
        """The parameters the engine trains.  `sft_modules` = the driver's `--sft_modules` (train_ds_medplib.py:54,316-326: substring
        families): of the fp32 tail exactly the named families train — `text_hidden_fcs` and / or `mask_decoder`, nothing
        unconditionally; the decoder-side families (lm_head, embed_tokens, norm weights, wg, front-end modules) train through the
        decoder backward, i.e. they must have been handed to `enable_lora(..., sft_modules=)` first — naming one without that state is
        an error, not a silently smaller training set.  Without the argument: the stage-III selection (text_hidden_fcs + mask decoder
        when config.train_mask_decoder) as before.  The LoRA adapters come along when enable_lora() attached them (get_peft_model
        marks exactly those trainable, :294-303)."""
        lora = getattr(self.model, "lora", None)
        if sft_modules is None:
            fams = ["text_hidden_fcs"] + (["mask_decoder"] if self.config.train_mask_decoder else [])
        else:
            fams = [f for f in (sft_modules.split(",") if isinstance(sft_modules, str) else list(sft_modules)) if f]
            unknown = [f for f in fams if f not in self.TAIL_FAMILIES + self.DECODER_SIDE_FAMILIES]
            if unknown:
                raise ValueError(f"--sft_modules {unknown}: not a trainable family of this build "
                                 f"({', '.join(self.TAIL_FAMILIES + self.DECODER_SIDE_FAMILIES)})")
            needs_decoder = [f for f in fams if f in self.DECODER_SIDE_FAMILIES]
            if needs_decoder and lora is None:
                raise ValueError(f"--sft_modules {needs_decoder} train through the decoder backward: call enable_lora(..., sft_modules=...) "
                                 "first (train.py: --lora_r > 0), or drive the model.MedPLIB surface, whose resolve_training_plan builds that state")
        ps = []
        for fam, mod in (("text_hidden_fcs", self.model.text_hidden_fcs), ("mask_decoder", self.model.visual_model.mask_decoder)):
            chosen = fam in fams
            for p in mod.parameters():
                if sft_modules is not None:
                    p.requires_grad_(chosen)         # an unnamed tail family is FROZEN (no gradient work, not in checkpoints), as in the reference
            if chosen:
                ps += list(mod.parameters())
        if lora is not None:
            ps += list(lora.parameters())
        if not ps:
            raise ValueError("nothing to train: --sft_modules is empty and no adapters are attached")
        return ps

    def _require_merged(self, what):
        if getattr(self.model, "lora", None) is not None:
            raise RuntimeError(f"{what}() works on the plain weights (KV-cache decode, export): call merge_and_unload() first (the "
                               "reference merges its adapters before inference too, merge_lora_weights_and_save_hf_model_moe.py)")

    def merge_and_unload(self):
        """peft `merge_and_unload()`: fold the trained adapters into the weights and drop them; inference / evaluate() / export then see
        the fine-tuned model (the adapters only act in the training forward)."""
        if getattr(self.model, "lora", None) is not None:
            self.model.lora.merge_into(self.model.llm)
            self.model.lora = None
            self.model.llm.lora = None
        return self

    def enable_lora(self, lora_r=8, lora_alpha=16, lora_dropout=0.0, lora_target_modules=("gate_proj", "up_proj", "down_proj"), seed=0,
                    train_gate=True, sft_modules=()):
        """get_peft_model(LoraConfig(r, lora_alpha, target_modules, lora_dropout)) for the decoder's MLP projections
        (train_ds_medplib.py:262-303; scripts/train_stage3.sh).  Call after the weights are loaded."""
        from . import llama_lora as LL
        targets = tuple(t for t in (lora_target_modules.split(",") if isinstance(lora_target_modules, str) else lora_target_modules))
        sft = tuple(x for x in (sft_modules.split(",") if isinstance(sft_modules, str) else sft_modules) if x in ("lm_head", "embed_tokens", "input_layernorm", "post_attention_layernorm"))
        self.model.lora = LL.enable_lora(self.model.llm, self.config, lora_r, lora_alpha, lora_dropout, targets, seed,
                                         train_gate and ("wg" in sft_modules or sft_modules == ()), sft)
        if "mask_encoder" in sft_modules and self.model.mask_encoder is not None:
            self.model.lora.add_mask_encoder(self.model.mask_encoder)
        if "region_fea_adapter" in sft_modules:
            self.model.lora.add_region_adapter(self.model.vision_tower)
        if "mm_token_compressor" in sft_modules and self.model.mm_token_compressor is not None:
            self.model.lora.add_token_compressor(self.model.mm_token_compressor)
        if "mm_projector" in sft_modules:
            self.model.lora.add_projector(self.model.vision_tower)
            self._want_raw_feats = True
        return self.model.lora

    def train(self, mode=True):
        super().train(mode)
        self.model.llm.training = mode
        return self

    # ------------------------------------------------------------------ checkpoint layout
    def load_hf_state_dict(self, sd):
        self.sync_side_streams()
        m = self.model
        m.llm.load_hf(sd)
        m.vision_tower.load_hf(sd)
        vm = {k[len("model.visual_model."):]: v for k, v in sd.items() if k.startswith("model.visual_model.")}
        self.load_sam_state_dict(vm)
        fc = {k[len("model.text_hidden_fcs."):]: v for k, v in sd.items() if k.startswith("model.text_hidden_fcs.")}
        m.text_hidden_fcs.load_state_dict({k: v.float() for k, v in fc.items()})
        if m.mm_token_compressor is not None:
            m.mm_token_compressor.load_hf(sd)
        if m.mask_encoder is not None:
            m.mask_encoder.load_hf(sd)

    def hf_state_dict(self):
        """Every weight in the reference's checkpoint key layout (what `model.state_dict()` of the reference class holds after
        `merge_and_unload()`, merge_lora_weights_and_save_hf_model_moe.py:338-343): `model.layers.*` / `lm_head` /
        `model.vision_tower.*` / `model.mm_projector.*` / `model.visual_model.*` / `model.text_hidden_fcs.*` (+ the ICL modules and
        the region adapter).  Adapters must be merged first.  `load_hf_state_dict(hf_state_dict())` is the identity."""
        self.sync_side_streams()
        self._require_merged("hf_state_dict")
        m = self.model
        sd = dict(m.llm.export_hf())
        sd.update(m.vision_tower.export_hf())
        vm = m.visual_model
        sd.update(vm.image_encoder.export_ref("model.visual_model.image_encoder."))
        for name, mod in (("mask_decoder", vm.mask_decoder), ("prompt_encoder", vm.prompt_encoder)):
            sd.update({f"model.visual_model.{name}.{k}": v for k, v in mod.state_dict().items()})
        sd.update({"model.text_hidden_fcs." + k: v for k, v in m.text_hidden_fcs.state_dict().items()})
        if m.mm_token_compressor is not None:
            sd.update(m.mm_token_compressor.export_hf())
        if m.mask_encoder is not None:
            sd.update(m.mask_encoder.export_hf())
        return {k: v.detach() for k, v in sd.items()}

    def save_pretrained(self, save_path, state_dict=None):
        """`<save_path>/pytorch_model.bin` (one torch.save file, tensors on the host) + `config.json` with this build's config
        fields -- the hand-over to the reference's inference scripts after training / merging here."""
        import dataclasses
        import json
        os.makedirs(save_path, exist_ok=True)
        sd = state_dict if state_dict is not None else self.hf_state_dict()
        torch.save({k: v.detach().cpu() for k, v in sd.items()}, os.path.join(save_path, "pytorch_model.bin"))
        cfg = dataclasses.asdict(self.config) if dataclasses.is_dataclass(self.config) else dict(vars(self.config))
        json.dump({k: (list(v) if isinstance(v, (tuple, set)) else v) for k, v in cfg.items()}, open(os.path.join(save_path, "config.json"), "w"),
                  indent=1, default=str)

    def load_sam_state_dict(self, sd):
        """`torch.load(path)['model']` key layout of SAM-Med2D checkpoints (build_sam.py:123-128), loaded non-strict."""
        vm = self.model.visual_model
        vm.image_encoder.load_ref(sd, "image_encoder.")
        dec = {k[len("mask_decoder."):]: v.float() for k, v in sd.items() if k.startswith("mask_decoder.")}
        vm.mask_decoder.load_state_dict(dec)
        pe = {k[len("prompt_encoder."):]: v.float() for k, v in sd.items() if k.startswith("prompt_encoder.")}
        vm.prompt_encoder.load_state_dict(pe, strict=False)
        vm.prompt_encoder._pe_cache = None

    # ------------------------------------------------------------------ pieces of model_forward
    def get_visual_embs(self, images):
        """get_visual_embs (MedPLIB.py:274-285): whole batch in one pass, tokens [B, 256, 256] bf16 (NHWC order)."""
        return self.model.visual_model.image_encoder.forward(images)

    @staticmethod
    def expand_index(valid_mask_bool, batch):
        """expand_embedding (MedPLIB.py:292-308) as a row index list: image i repeated once per mask of sample i."""
        if valid_mask_bool is None or len(valid_mask_bool) == 0:
            return list(range(batch))
        idx = []
        for i, m in enumerate(valid_mask_bool):
            idx += [i] * (len(m) if m else 0)
        return idx

    def _postprocess(self, low_res, resize_list, shapes):
        """postprocess_masks (MedPLIB.py:682-701) grouped by (input_size, original_size) so equal shapes share a launch."""
        n = low_res.shape[0]
        groups = {}
        for i in range(n):
            groups.setdefault((tuple(int(v) for v in resize_list[i]), tuple(int(v) for v in shapes[i])), []).append(i)
        outs = [None] * n
        if len(groups) == 1:
            (inp, orig), _ = next(iter(groups.items()))
            crop = ops.postprocess_crop(low_res.shape[-2], low_res.shape[-1], inp)
            full = A.BilinearResizeFn.apply(low_res, crop, orig)
            return full, [full[i:i + 1] for i in range(n)]
        for (inp, orig), ids in groups.items():
            crop = ops.postprocess_crop(low_res.shape[-2], low_res.shape[-1], inp)
            sel = low_res[ids] if len(ids) != n else low_res
            r = A.BilinearResizeFn.apply(sel, crop, orig)
            for j, i in enumerate(ids):
                outs[i] = r[j:j + 1]
        return None, outs

    def _region_features(self, raw_feats, n_images, region_masks, valid_region_masks_bool):
        """extract_region_feature (medplib_arch.py:283-295, 580-613) for the samples that carry region masks: region_fea_adapter on
        their raw tower features, then per mask the mean of the feature map sampled at (a random subset of) its non-zero pixels.
        The pixel lists are built on the host (`nonzero` has a data-dependent size; the reference synchronises here too) and the
        subset is drawn with torch.randperm from the global generator exactly where the reference draws it.
        Returns (features [n_region_masks, hidden] bf16, per-sample first row or None)."""
        cfg = self.config
        valid = [any(v) for v in valid_region_masks_bool]
        vidx = [i for i, v in enumerate(valid) if v]
        assert len(vidx) == len(region_masks), f"{len(vidx)}, {len(region_masks)}"              # medplib_arch.py:582
        NP, C = cfg.clip_num_patches, cfg.clip_hidden_size
        hw = int(math.isqrt(NP))
        raw = raw_feats.view(n_images, NP, C)
        sel = raw if vidx == list(range(n_images)) else raw[torch.as_tensor(vidx, device=raw.device)]
        fmap = self.model.vision_tower.region_feature_map(sel.reshape(-1, C)).view(len(vidx), NP, cfg.hidden_size)
        xy, offsets, map_index, bases, n_rows = [], [0], [], [None] * n_images, 0
        for j, (b, masks) in enumerate(zip(vidx, region_masks)):
            if len(masks) == 0:
                continue
            bases[b] = n_rows
            H, W = masks[0].shape[0], masks[0].shape[1]
            for mk in masks:
                pts = torch.as_tensor(mk).detach().cpu().nonzero()                                  # (y, x)
                if pts.shape[0] > cfg.max_sample_point:
                    pts = pts[torch.randperm(pts.shape[0])[:cfg.max_sample_point], :]               # rand_sample, :33-39
                p = pts.to(torch.float32) / torch.tensor([float(H), float(W)])
                xy.append(p.flip(1).numpy())                                                        # (x, y) for grid_sample
                offsets.append(offsets[-1] + pts.shape[0])
                map_index.append(j)
                n_rows += 1
        dev = self.device_
        xy_d, off_d = _h2d(np.concatenate(xy).astype(np.float32).reshape(-1, 2), dev), _h2d(np.asarray(offsets, dtype=np.int64), dev)
        mi_d = _h2d(np.asarray(map_index, dtype=np.int32), dev)
        self._region_ctx = (sel.reshape(-1, C), xy_d, off_d, mi_d, hw)      # kept for a trainable region_fea_adapter
        feats = ops.region_point_mean(fmap.contiguous(), xy_d, off_d, mi_d, hw, hw)
        return feats, bases

    def _encode_and_plan(self, ids_np, lab_np, att_np, images_clip, mask_images=None, image_token_types=None,
                         image_token_lengths=None, with_seg=True, region_masks=None, valid_region_masks_bool=None):
        """encode_images (+ TokenCompressor) / encode_masks and the splice plan for the three image layouts of
        prepare_inputs_labels_for_multimodal (medplib_arch.py:246-279): 4-D tensor = one image per sample; list / 5-D =
        several images per sample; list + mask_images + image_token_types = ICL separate mode with the mask encoder.
        Returns (plan, feature rows [n_rows, hidden] bf16)."""
        cfg, m = self.config, self.model
        tok = cfg.image_token_len
        seg_idx = self.seg_token_idx if with_seg else None
        seg_lens = image_token_lengths if image_token_lengths is not None else tok
        multi = isinstance(images_clip, (list, tuple)) or images_clip.dim() == 5
        clip_in = torch.cat([im for im in images_clip], 0) if multi else images_clip
        region_flag = region_masks is not None and len(region_masks) > 0                        # medplib_arch.py:221-227
        self._last_raw = None
        if region_flag:
            assert not multi, "region prompts come with one image per sample"                     # medplib_arch.py:248, 269
            feats, raw = m.vision_tower.encode_images(clip_in, return_raw=True)
        elif getattr(self, "_want_raw_feats", False):                # a trainable mm_projector re-runs the projector from these
            feats, self._last_raw = m.vision_tower.encode_images(clip_in, return_raw=True)
        else:
            feats = m.vision_tower.encode_images(clip_in)
        if m.mm_token_compressor is not None:
            self._comp_in = (feats, clip_in.shape[0])               # kept for a trainable compressor (re-run on the autograd tape)
            feats = m.mm_token_compressor.forward(feats, clip_in.shape[0], cfg.clip_num_patches)
        if region_flag:
            rfeats, rbases = self._region_features(raw, clip_in.shape[0], region_masks, valid_region_masks_bool)
            n_img_rows = feats.shape[0]
            plan = plan_splice(ids_np, lab_np, att_np, tok, seg_token_idx=seg_idx, seg_feature_lengths=seg_lens,
                               region_bases=[None if b is None else n_img_rows + b for b in rbases])
            return plan, torch.cat([feats, rfeats], 0)
        if image_token_types is not None and mask_images is not None and len(mask_images) > 0:
            assert multi, "ICL separate mode expects a list (or 5-D tensor) of images"     # medplib_arch.py:247
            if m.mask_encoder is None:
                raise ValueError("mask_images given but the model was built without icl_mask_encoder")
            self._mask_in = torch.cat([mk for mk in mask_images], 0).to(self.device_)      # kept for a trainable mask encoder
            mfeats = m.mask_encoder.forward(self._mask_in)
            lengths, bases = icl_feature_layout(image_token_types, tok, cfg.mask_encoder_token_count)
            plan = plan_splice(ids_np, lab_np, att_np, lengths, seg_token_idx=seg_idx, seg_feature_lengths=seg_lens, feature_bases=bases)
            feats = torch.cat([feats, mfeats], 0)
        elif multi:
            plan = plan_splice(ids_np, lab_np, att_np, [tok] * clip_in.shape[0], seg_token_idx=seg_idx, seg_feature_lengths=seg_lens)
        else:
            plan = plan_splice(ids_np, lab_np, att_np, tok, seg_token_idx=seg_idx, seg_feature_lengths=seg_lens)
        return plan, feats

    # ------------------------------------------------------------------ forward
    def forward(self, **kwargs):
        return self.model_forward(**kwargs)

    def _lora_training_forward(self, plan, feats, src, embeds, key_valid, sup_rows_d, sup_labels_d, region_masks, mask_images, B):
        """LoRA / --sft_modules training (llama_lora.py): the decoder, the CE and the <SEG>-row gather are autograd Functions, so
        loss.backward() runs the whole decoder backward and leaves every trainable tensor's gradient in the engine's flat buffer.
        Front-end modules that train (projector, token compressor, mask encoder, region adapter) are re-run on the autograd tape
        from the frozen tensors the no-grad pass kept, and their rows replace the frozen ones in the feature block.
        -> (last_hidden [B, S, d] with grad, ce [1] with grad)."""
        from . import llama_lora as LL
        cfg, m, dev = self.config, self.model, self.device_
        lo = m.llm.lora
        decoder_params = [lo.params[lo.index[n_]] for n_ in LL.decoder_param_names(lo)]
        feats_on_tape = False
        proj_p = [lo.full_param(f"model.mm_projector.{k}") for k in ("0.weight", "0.bias", "2.weight", "2.bias")]
        if proj_p[0] is not None:                                   # mm_projector trains (stage II)
            assert self._last_raw is not None and feats.shape[0] == self._last_raw.shape[0], \
                "a trainable mm_projector is built for the plain image layout (no compressor / ICL / region rows)"
            feats, feats_on_tape = LL.ProjectorFn.apply(self._last_raw, m.vision_tower, *proj_p), True
        reg_p = [lo.full_param(f"model.region_fea_adapter.{k}") for k in ("weight", "bias")]
        if reg_p[0] is not None and region_masks is not None and len(region_masks) > 0:          # region_fea_adapter trains (stage IV)
            rsel, xy_d, off_d, mi_d, hw_ = self._region_ctx
            rnew = LL.RegionAdapterFn.apply(rsel, m.vision_tower, xy_d, off_d, mi_d, hw_, *reg_p)
            feats, feats_on_tape = torch.cat([feats[:feats.shape[0] - rnew.shape[0]], rnew], 0), True   # region rows sit behind the image rows
        menc_names = [n_ for n_ in lo.names if n_.startswith("model.mask_encoder.")]
        if menc_names and mask_images is not None and len(mask_images) > 0:                      # mask_encoder trains (ICL)
            mnew = LL.MaskEncoderFn.apply(self._mask_in, m.mask_encoder, *[lo.full_param(n_) for n_ in menc_names])
            feats, feats_on_tape = torch.cat([feats[:feats.shape[0] - mnew.shape[0]], mnew], 0), True   # mask rows sit behind the image rows
        comp_p = [lo.full_param(f"model.mm_token_compressor.{k}") for k in ("norm.weight", "norm.bias", "proj.weight", "proj.bias")]
        if comp_p[0] is not None:                                   # mm_token_compressor trains (train_medplib_icl.sh)
            cin, n_img = self._comp_in
            new = LL.TokenCompressorFn.apply(cin, m.mm_token_compressor, n_img, cfg.clip_num_patches, *comp_p)
            feats = new if feats.shape[0] == new.shape[0] else torch.cat([new, feats[new.shape[0]:]], 0)   # + mask / region rows
            feats_on_tape = True
        emb_p = lo.full_param("model.embed_tokens.weight")
        if emb_p is not None or feats_on_tape:                      # the splice joins the autograd tape
            embeds = LL.EmbedSpliceFn.apply(emb_p, m.llm, feats, src, plan.src_code, (B, plan.seq_len, cfg.hidden_size))
        last_hidden, aux_sum = LL.LlamaLoRAFn.apply(m.llm, embeds, key_valid, *decoder_params)
        if sup_rows_d.numel():
            ce = LL.CrossEntropyFn.apply(last_hidden, sup_rows_d, sup_labels_d, m.llm, lo.full_param("lm_head.weight"))
        else:
            ce = torch.full((1,), float("nan"), dtype=torch.float32, device=dev)
        if m.llm.moe_layers and cfg.router_aux_loss_coef != 0.0:
            ce = ce + cfg.router_aux_loss_coef * aux_sum           # medplib_moe_llama.py:410-421
        return last_hidden, ce

    def model_forward(self, images, images_clip, input_ids, labels, attention_mask=None, masks_list=None, label_list=None,
                      resize_list=None, region_masks=None, offset=None, inference=False, seg_flag=True, valid_mask_bool=None,
                      rp_flag=False, valid_region_masks_bool=None, attention_masks=None, **kwargs):
        cfg = self.config
        if attention_mask is None:
            attention_mask = attention_masks            # LISA.model_forward spells it `attention_masks` (LISA.py:267)
        dev = self.device_
        ids_np = _np_ids(input_ids)
        lab_np = _np_ids(labels) if labels is not None else None
        att_np = _np_ids(attention_mask).astype(bool) if attention_mask is not None else None
        B = ids_np.shape[0]
        m = self.model
        # The frozen SAM-Med2D encoder does not depend on the LLM: it runs on a side stream so its ~350 short, low-occupancy
        # launches (windowed attention, adapter convolutions on 64-token maps) fill in beside the LLM's GEMMs instead of
        # occupying the machine alone.  Joined again before the mask tail.
        image_tokens = None
        lo_ = getattr(m.llm, "lora", None)
        front_frozen = lo_ is None or not any(n.startswith(("model.mm_projector.", "model.mm_token_compressor.", "model.region_fea_adapter.",
                                                             "model.mask_encoder.")) for n in lo_.names)
        ahead = (self.towers_run_ahead and front_frozen and not getattr(self, "_want_raw_feats", False)
                 and not (region_masks is not None and len(region_masks) > 0) and kwargs.get("mask_images") is None and torch.is_tensor(images_clip))
        # Towers that run ahead are issued BEFORE this step's decoder, i.e. while the device is still inside the PREVIOUS step's decoder: with
        # gate_towers on, layer j of the CLIP tower / block k of the SAM encoder waits for the event the previous step's decoder recorded where
        # its layer j / 32 - 12 + k started its expert GEMMs (llama._mlp) — the gate|up launch (5.71 waves of tiles: dynamic, with slack in its last
        # wave) absorbs side workgroups, the exact-wave qkv / o_proj launches (768 / 256 tiles on 256 CUs) end late by whatever a side workgroup
        # still holds a CU for at their start.  Events already complete (first step, a host that is not ahead) gate nothing.
        n_l = len(m.llm.layers)
        gate_on = bool(ahead and getattr(self, "gate_towers", _GATE_TOWERS) and cfg.moe_enable and n_l >= 24 and self.training and not inference)
        if gate_on and getattr(m.llm, "layer_events", None) is None:
            m.llm.layer_events = [torch.cuda.Event() for _ in range(n_l)]
            for e_ in m.llm.layer_events:
                e_.record()                                  # so that the first step's waits see a completed record
        enc = m.visual_model.image_encoder if seg_flag else None
        if seg_flag and self.sam_side_stream:
            main = torch.cuda.current_stream()
            side = self._side_stream()
            if not ahead:
                side.wait_stream(main)
            else:
                images.record_stream(side)
            enc.gate_events = m.llm.layer_events[n_l - len(enc.blocks):] if (gate_on and _GATE_MODE != 2) else None
            with torch.cuda.stream(side), torch.no_grad():
                image_tokens = ops.cast_to_f32(self.get_visual_embs(images))
            enc.gate_events = None
        if ahead:
            main, vis = torch.cuda.current_stream(), self._vision_stream()
            images_clip.record_stream(vis)
            m.vision_tower.gate_events = m.llm.layer_events if (gate_on and _GATE_MODE != 2) else None
            with torch.cuda.stream(vis), torch.no_grad():
                plan, feats = self._encode_and_plan(ids_np, lab_np, att_np, images_clip, None, kwargs.get("image_token_types"),
                                                    kwargs.get("image_token_lengths"))
            m.vision_tower.gate_events = None
            main.wait_stream(vis)
            feats.record_stream(main)
        with torch.no_grad():
            if not ahead:
                plan, feats = self._encode_and_plan(ids_np, lab_np, att_np, images_clip, kwargs.get("mask_images"),
                                                    kwargs.get("image_token_types"), kwargs.get("image_token_lengths"),
                                                    region_masks=region_masks, valid_region_masks_bool=valid_region_masks_bool)
            # every host-built index tensor of the step goes to the device NOW (see _h2d)
            src = _h2d(plan.src_code.reshape(-1), dev)
            key_valid = None
            if plan.attention_mask is not None and not plan.attention_mask.all():
                key_valid = _h2d(plan.attention_mask.astype(np.uint8), dev)
            sup_rows, sup_labels = plan.supervised() if lab_np is not None else (np.zeros(0, np.int64),) * 2
            sup_rows_d, sup_labels_d = _h2d(sup_rows, dev), _h2d(sup_labels, dev)
            seg_rows = plan.seg_rows()
            n_masks_given = len(masks_list) if masks_list is not None else 0
            if kwargs.get("icl_image_counts") is not None and n_masks_given > 0:
                seg_rows = seg_rows[-n_masks_given:]                   # MedPLIB.py:462-463
            seg_rows_d = _h2d(seg_rows, dev) if seg_flag else None
            # The rows of the decoder's output that anything reads: the supervised rows of the filtered CE and the <SEG> rows.  The LAST layer's
            # MLP is row-wise, so it runs on those rows only (unread rows are not computed; every read row has the bits it
            # would have had).  Off while a test captures the whole hidden state, at inference, or by MP_PRUNE_LAST_MLP=0.
            m.llm.needed_rows = None
            m.llm.pruned_rows = None
            if (_PRUNE_LAST_MLP and self.training and not inference and not self.capture_intermediates
                    and kwargs.get("icl_image_counts") is None):
                need = np.union1d(np.asarray(sup_rows, dtype=np.int64), np.asarray(seg_rows if seg_flag else [], dtype=np.int64))
                T_all = B * plan.seq_len
                if 0 < need.size < T_all // 2:
                    mask = np.zeros(T_all, dtype=np.uint8)
                    mask[need] = 1
                    m.llm.needed_rows = (_h2d(need, dev), _h2d(mask, dev))
            exp = self.expand_index(valid_mask_bool, B) if seg_flag else None
            exp_d = _h2d(np.asarray(exp, dtype=np.int64), dev) if (seg_flag and exp != list(range(B))) else None
            if getattr(m.llm, "lora", None) is not None:
                m.llm.lora.sync_model(m.llm)                        # bf16 working copies of lm_head / embed_tokens when they train
            embeds = ops.splice_rows(m.llm.embed_tokens, feats, src, cfg.hidden_size).view(B, plan.seq_len, cfg.hidden_size)
            # adapters attached: they act in every forward of this method (training, validation, inference masks) like a peft
            # model; the KV-cache decode paths want them merged (merge_and_unload())
            lora_train = getattr(m.llm, "lora", None) is not None
            if not lora_train:
                last_hidden, aux, self._routing = m.llm.forward(embeds, key_valid, collect_routing=self.capture_intermediates)
                ce = m.llm.cross_entropy(last_hidden, sup_rows_d, sup_labels_d, aux)
        if lora_train and torch.is_grad_enabled() and self.training and not inference:
            last_hidden, ce = self._lora_training_forward(plan, feats, src, embeds, key_valid, sup_rows_d, sup_labels_d, region_masks,
                                                          kwargs.get("mask_images"), B)
        elif lora_train:
            from . import llama_lora as LL
            with torch.no_grad():
                last_hidden, aux_sum, _ = LL.forward_train(m.llm, embeds, key_valid)
                ce = m.llm.cross_entropy(last_hidden, sup_rows_d, sup_labels_d, [aux_sum] if m.llm.moe_layers else [])
        # the row set belonged to THIS pass through the stack: a later direct call of the stack (evaluate(), a test) computes every row
        nr, done = m.llm.needed_rows, getattr(m.llm, "pruned_rows", None)
        # (rows the last layer's MLP ran on, rows of the batch) when the stack really pruned (top-1 fused-gate MoE branch / dense last layer with
        # frozen ln2 under adapters: it records llm.pruned_rows), else None — bench.py reports and subtracts FLOPs only from this
        self.last_pruned = (int(done), int(nr[1].numel())) if (nr is not None and done is not None) else None
        m.llm.needed_rows = None
        if not seg_flag:
            z = torch.zeros(1, dtype=torch.float32, device=dev)
            ce_w = ce * cfg.ce_loss_weight if ce.requires_grad else ops.mean_plus(ce, cfg.ce_loss_weight)          # ce * ce_loss_weight
            out = {k: z[0] for k in LOSS_KEYS}
            out["loss"] = out["ce_loss"] = ce_w[0]
            return out

        # ---- <SEG> rows -> text_hidden_fcs -> prompt + mask decoder -> losses (MedPLIB.py:456-572), batched over all masks
        main = torch.cuda.current_stream()
        if image_tokens is None:
            with torch.no_grad():
                image_tokens = ops.cast_to_f32(self.get_visual_embs(images))
        else:
            main.wait_stream(self._side_stream())
            image_tokens.record_stream(main)
        use_tail = self.tail_side_stream and self.training and not inference and torch.is_grad_enabled()
        m.visual_model.mask_decoder.program_grid = 64 if use_tail else 256       # hidden beside the decoder: fewer resident workgroups (sam.py)
        if not use_tail:
            self.active_tail_stream = None
            if self._tail_stream_obj is not None:
                main.wait_stream(self._tail_stream_obj)           # parameters may still be in an optimizer step over there
            return self._mask_tail(plan, last_hidden, ce, image_tokens, seg_rows_d, exp_d, masks_list, label_list, resize_list,
                                   inference, B)
        tail = self._tail_stream()
        tail.wait_stream(main)
        for t in (last_hidden, ce, image_tokens, seg_rows_d, exp_d):
            if t is not None:
                t.record_stream(tail)
        for g_ in (masks_list or []):
            if torch.is_tensor(g_) and g_.is_cuda:
                g_.record_stream(tail)
        with torch.cuda.stream(tail):
            out = self._mask_tail(plan, last_hidden, ce, image_tokens, seg_rows_d, exp_d, masks_list, label_list, resize_list,
                                  inference, B)
        # The loss tensors are produced on the tail stream.  Nothing on the caller's stream needs them unless the caller reads one, so the
        # cross-stream wait is attached to the READ (StreamOrderedLosses) instead of being paid by the next step's CLIP tower and decoder:
        # engine.backward(out) takes the dict as it is; out["loss"] / float(out["loss"]) / the reference driver's loss.item() wait first.
        self.active_tail_stream = tail
        res = StreamOrderedLosses(out, tail.record_event())
        if os.environ.get("MP_TAIL_WAIT") == "1":              # A/B: the calling stream waits here, as before round 2
            res._order()
        return res

    def _mask_tail(self, plan, last_hidden, ce, image_tokens, seg_rows_d, exp_d, masks_list, label_list, resize_list, inference, B):
        cfg, dev, m = self.config, self.device_, self.model
        with torch.no_grad():
            if self.capture_intermediates:
                self.captured = {"last_hidden": last_hidden, "image_tokens": image_tokens, "ce": ce,
                                 "routing": getattr(self, "_routing", None)}
            assert image_tokens.shape[0] == B
            if exp_d is not None:
                image_tokens = ops.gather_rows_f32(image_tokens, exp_d)
            hidden_rows = None if last_hidden.requires_grad else ops.gather_rows_bf16_to_f32(last_hidden.view(-1, cfg.hidden_size), seg_rows_d)
        if hidden_rows is None:                                   # LoRA training: the <SEG> rows carry gradient back into the decoder
            from . import llama_lora as LL
            hidden_rows = LL.GatherRowsFn.apply(last_hidden, seg_rows_d)
        n = hidden_rows.shape[0]
        assert n <= image_tokens.shape[0], "more <SEG> rows than expanded image embeddings"   # pairing by position, :473-487
        if n < image_tokens.shape[0]:
            image_tokens = image_tokens[:n].contiguous()
        fc = m.text_hidden_fcs[0]
        pe = m.visual_model.prompt_encoder
        # text_hidden_fcs (MedPLIB.py:152-164) runs inside the mask decoder's call: with the tail program the whole trainable chain up to the
        # upsampler is one launch forward and one backward
        low_res, iou_pred = m.visual_model.mask_decoder(image_tokens, pe.dense_pe_tokens(), pe.no_mask_embed.weight, None,
                                                         fcs=(fc[0], fc[2]), hidden_rows=hidden_rows)
        shapes = [tuple(l.shape[-2:]) for l in label_list[:n]]
        full, pred_masks = self._postprocess(low_res, resize_list[:n], shapes)
        if inference:
            return {"pred_masks": pred_masks, "gt_masks": masks_list}

        # ---- fused mask losses (MedPLIB.py:515-572)
        weights = (cfg.ce_loss_weight, cfg.bce_loss_weight, cfg.dice_loss_weight, cfg.iou_loss_weight, cfg.focal_loss_weight)
        if full is not None:
            H, W = full.shape[-2:]
            gt = torch.stack([g.reshape(H, W) for g in masks_list[:n]]).to(device=dev, dtype=torch.float32).view(n, H * W)
            out10 = A.MaskLossFn.apply(full.view(n, H * W), gt, iou_pred, ce, weights)
        else:
            # masks of different sizes in one step: one ragged launch over the flat concatenation (the reference loops per mask)
            for i in range(n):
                assert tuple(masks_list[i].shape[-2:]) == tuple(pred_masks[i].shape[-2:]), \
                    f"gt_mask.shape: {tuple(masks_list[i].shape)}, pred_mask.shape: {tuple(pred_masks[i].shape)}"   # MedPLIB.py:527-531
            sizes = [int(pm.shape[-2] * pm.shape[-1]) for pm in pred_masks]
            offsets = _h2d(np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64), dev)
            pred_flat = torch.cat([pm.reshape(-1) for pm in pred_masks])
            gt_flat = torch.cat([g.reshape(-1).to(device=dev, dtype=torch.float32) for g in masks_list[:n]])
            out10 = A.MaskLossFn.apply(pred_flat, gt_flat, iou_pred, ce, weights, offsets)
        return {k: out10[i] for i, k in enumerate(LOSS_KEYS)}


    def _decode_graph(self, prefill_hidden, cache, S, max_new_tokens, eos_token_id, check_every=16):
        """Greedy decode with ONE captured HIP graph per call: embed(token) -> 32 decode layers (GEMV projections, RoPE + KV append
        and attention at the device-side cache length) -> final norm -> lm_head -> argmax -> next token, all on the device; the host
        replays the graph once per token (~450 kernel launches otherwise) and only looks at the generated ids every `check_every`
        tokens to stop at EOS.  Tokens after the first EOS are discarded, so the result equals the token-by-token loop.
        Returns (generated ids, [hidden of each fed token])."""
        cfg, dev, llm = self.config, self.device_, self.model.llm
        d = cfg.hidden_size
        tok = ops.argmax_rows(llm.next_token_logits(prefill_hidden[0, -1:]))              # first generated token, int64 [1] on the device
        counters = torch.tensor([S, S + 1], dtype=torch.int32, device=dev)
        toks = torch.empty(max_new_tokens, dtype=torch.int64, device=dev)
        hid_all = torch.empty((max_new_tokens, d), dtype=torch.bfloat16, device=dev)
        toks[0:1].copy_(tok)
        h_static = torch.empty((1, 1, d), dtype=torch.bfloat16, device=dev)

        def step():
            emb = ops.splice_rows(llm.embed_tokens, None, tok, d)
            h = llm.decode_step(emb.view(1, 1, d), cache, counters)
            h_static.copy_(h)
            tok.copy_(ops.argmax_rows(llm.next_token_logits(h[0, -1:])))
            ops.advance_ints(counters, 1)

        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream())
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.stream(side):
            with torch.cuda.graph(graph, stream=side):
                step()
        torch.cuda.current_stream().wait_stream(side)
        n, done = 1, False
        while n < max_new_tokens and not done:
            upto = min(max_new_tokens, n + check_every)
            for i in range(n, upto):
                graph.replay()                                   # feeds token i-1, produces token i
                hid_all[i - 1:i].copy_(h_static.view(1, d))
                toks[i:i + 1].copy_(tok)
            got = toks[:upto].cpu().tolist()                     # the only host synchronisation of the loop
            n = upto
            if eos_token_id in got:
                n = got.index(eos_token_id) + 1
                done = True
        generated = toks[:n].cpu().tolist()
        step_hiddens = [hid_all[i:i + 1].view(1, 1, d) for i in range(n - 1)]
        return generated, step_hiddens

    def _greedy(self, ids, images_clip, max_new_tokens, eos_token_id, mask_images=None, image_token_types=None, image_token_lengths=None,
                region_masks=None, valid_region_masks_bool=None):
        """HF `generate(do_sample=False, use_cache=True)` on one sample (MedPLIB.py:592-606; prepare_inputs_for_generation,
        medplib_moe_llama.py:451-485): prefill of the spliced prompt, then single-token decode steps against the KV cache.
        Returns (output_ids [1, L + n] on the host, [hidden states of the prompt, of each fed token])."""
        cfg, dev, m = self.config, self.device_, self.model
        plan, feats = self._encode_and_plan(ids, None, None, images_clip, mask_images, image_token_types, image_token_lengths,
                                            with_seg=False, region_masks=region_masks if region_masks else None,
                                            valid_region_masks_bool=valid_region_masks_bool)
        src = _h2d(plan.src_code.reshape(-1), dev)
        S = plan.seq_len
        embeds = ops.splice_rows(m.llm.embed_tokens, feats, src, cfg.hidden_size).view(1, S, cfg.hidden_size)
        cache = m.llm.new_kv_cache(1, S + max_new_tokens)
        hidden, _, _ = m.llm.forward(embeds, None, kv_cache=cache)
        if self.decode_with_graph and max_new_tokens > 2 and cfg.top_k_experts == 1:
            generated, step_hiddens = self._decode_graph(hidden, cache, S, max_new_tokens, eos_token_id)
        else:
            generated, step_hiddens = [], []
            last = hidden[0, -1:]
            for _ in range(max_new_tokens):
                logits = m.llm.next_token_logits(last)
                tok = int(ops.argmax_rows(logits)[0])
                generated.append(tok)
                if tok == eos_token_id or len(generated) == max_new_tokens:
                    break
                emb = ops.splice_rows(m.llm.embed_tokens, None, torch.tensor([tok], dtype=torch.int64, device=dev), cfg.hidden_size)
                h, _, _ = m.llm.forward(emb.view(1, 1, -1), None, kv_cache=cache)
                step_hiddens.append(h)
                last = h[0, -1:]
        return np.concatenate([ids, np.asarray(generated, dtype=np.int64)[None]], 1), [hidden] + step_hiddens

    @torch.no_grad()
    def generate(self, input_ids, images=None, attention_mask=None, max_new_tokens=512, eos_token_id=2, **kwargs):
        """The VQA entry the drivers use (`model.generate(input_ids, images=images_clip, attention_mask=..., max_new_tokens=...,
        do_sample=False)`, vqa_infer.py:430-442): greedy decoding, one sample per call like the reference's loop (batch rows are
        decoded one after the other).  Returns output ids [B, L + n_max] (right-padded with eos), prompt included, as HF does."""
        # HF generate kwargs the reference driver passes (vqa_infer.py:430-442): accepted when they mean greedy decoding, refused
        # loudly otherwise — sampling and beam search are not built
        if kwargs.get("do_sample") or (kwargs.get("temperature") or 0) > 0 and kwargs.get("do_sample") is not False:
            raise NotImplementedError("generate(): sampling (do_sample / temperature > 0) is not built; the shipped eval scripts decode greedily")
        if (kwargs.get("num_beams") or 1) != 1:
            raise NotImplementedError("generate(): beam search (num_beams > 1) is not built; the shipped eval scripts use num_beams=1")
        unknown = set(kwargs) - {"do_sample", "temperature", "top_p", "num_beams", "use_cache", "mask_images", "image_token_types",
                                 "image_token_lengths", "region_masks", "valid_region_masks_bool", "output_hidden_states",
                                 "return_dict_in_generate", "pad_token_id"}
        if unknown:
            raise TypeError(f"generate(): unsupported arguments {sorted(unknown)}")
        self.sync_side_streams()
        self._require_merged("generate")
        ids = _np_ids(input_ids).astype(np.int64)
        was_training = self.training
        self.train(False)
        rows = []
        for b in range(ids.shape[0]):
            row = ids[b:b + 1]
            if attention_mask is not None:
                row = row[:, _np_ids(attention_mask)[b].astype(bool)]               # drop this row's padding
            img = images[b] if isinstance(images, (list, tuple)) else images[b:b + 1]
            if isinstance(images, (list, tuple)):
                img = [img]
            rm, rv = kwargs.get("region_masks") or (), kwargs.get("valid_region_masks_bool") or ()
            if len(rm) > 0:            # the collator's flat list holds one entry per sample WITH regions: pick this row's
                before = sum(1 for v in rv[:b] if any(v))
                rm, rv = (rm[before:before + 1] if any(rv[b]) else ()), rv[b:b + 1]
            out, _ = self._greedy(row, img, max_new_tokens, eos_token_id, kwargs.get("mask_images"), kwargs.get("image_token_types"),
                                  kwargs.get("image_token_lengths"), rm, rv)
            rows.append(out[0])
        self.train(was_training)
        n = max(r.shape[0] for r in rows)
        return torch.from_numpy(np.stack([np.concatenate([r, np.full(n - r.shape[0], eos_token_id, np.int64)]) for r in rows]))

    # ------------------------------------------------------------------ evaluate (MedPLIB.py:574-680)
    @torch.no_grad()
    def evaluate(self, images_clip, images, input_ids, resize_list, original_size_list, region_masks=(), valid_region_masks_bool=(),
                 max_new_tokens=512, tokenizer=None, attention_mask=None, inference_demo=False, mask_images=None,
                 image_token_types=None, image_token_lengths=None, eos_token_id=2):
        """Greedy generation with a KV cache (prefill + single-token decode steps), then one mask per sample from the hidden
        state that predicts the first <SEG> (or position -2 when no <SEG> was generated) — MedPLIB.py:574-680.
        Returns (output_ids [1, L + n_generated] int64 on the host, [pred_mask [1,H,W]]).

        Reference quirk kept: the concatenated per-step hidden states cover the spliced prompt and the first n-1 generated
        tokens (the last generated token is never fed back), i.e. one position FEWER than build_seg_token_mask(output_ids)
        yields; the mask's final position is always False (shifted mask), so it is truncated to the hidden length."""
        cfg, dev, m = self.config, self.device_, self.model
        self.sync_side_streams()
        self._require_merged("evaluate")
        ids = _np_ids(input_ids).astype(np.int64)
        assert ids.shape[0] == 1, "evaluate() decodes one sample at a time, like the reference's validate_seg (vqa_infer.py:528)"
        was_training = self.training
        self.train(False)
        nfeat = cfg.image_token_len
        output_ids, hiddens = self._greedy(ids, images_clip, max_new_tokens, eos_token_id, mask_images, image_token_types, image_token_lengths,
                                           region_masks, valid_region_masks_bool)
        all_hidden = torch.cat(hiddens, 1)                                  # [1, S + n_gen - 1, d]
        n_hidden = all_hidden.shape[1]
        if (output_ids[:, 1:] == self.seg_token_idx).sum() == 0 and inference_demo:
            self.train(was_training)
            return torch.from_numpy(output_ids), []
        n_ph = int((output_ids == IMAGE_TOKEN_INDEX).sum())
        has_region = bool((output_ids == REGION_TOKEN_INDEX).any())          # region ids expand 1:1, any base will do for the mask
        seg_plan = plan_splice(output_ids, None, None, nfeat if n_ph <= 1 else [nfeat] * n_ph, seg_token_idx=self.seg_token_idx,
                               seg_feature_lengths=image_token_lengths if image_token_lengths is not None else nfeat,
                               region_bases=[0] * output_ids.shape[0] if has_region else None)
        seg_rows = np.flatnonzero(seg_plan.seg_mask[0, :n_hidden])
        if seg_rows.size >= 1:
            row = int(seg_rows[0])                                          # first <SEG> when several (MedPLIB.py:639-641)
        else:
            row = n_hidden - 2                                              # last_hidden_state[:1, -2:-1] (MedPLIB.py:642-644)
        hid = ops.gather_rows_bf16_to_f32(all_hidden.view(-1, cfg.hidden_size), torch.tensor([row], dtype=torch.int64, device=dev))
        fc = m.text_hidden_fcs[0]
        pred_emb = A.linear(A.linear(hid, fc[0].weight, fc[0].bias, ops.SACT_RELU), fc[2].weight, fc[2].bias)
        image_tokens = ops.cast_to_f32(self.get_visual_embs(images))[:1].contiguous()
        pe = m.visual_model.prompt_encoder
        low_res, _ = m.visual_model.mask_decoder(image_tokens, pe.dense_pe_tokens(), pe.no_mask_embed.weight, pred_emb.view(1, 1, -1))
        osz = original_size_list[0]
        shape = tuple(osz.shape[-2:]) if hasattr(osz, "shape") else tuple(osz)
        _, pred_masks = self._postprocess(low_res, resize_list[:1], [shape])
        self.train(was_training)
        return torch.from_numpy(output_ids), pred_masks


class LISAForCausalLM(MedPLIBForCausalLM):
    """Dense (non-MoE) twin (model/LISA.py:180-471): same path with plain LlamaMLP layers; accepts the collator's
    `attention_mask` as well as LISA's own `attention_masks` spelling."""
    moe_default = False
