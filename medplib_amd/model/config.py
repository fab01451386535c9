"""Model configuration: the numbers the reference reads from checkpoint `config.json`s (LLaVA-1.5-7B / Vicuna-7B,
CLIP ViT-L/14-336, SAM-Med2D ViT-B @256) plus the kwargs `MedPLIBForCausalLM.__init__` pops (model/MedPLIB.py:195-234)."""
from dataclasses import dataclass, field
from typing import List, Optional


@dataclass
class MedPLIBConfig:
    # Llama
    vocab_size: int = 32267                 # 32000 + 267 added tokens (train_ds_medplib.py:207-216)
    hidden_size: int = 4096
    intermediate_size: int = 11008
    num_hidden_layers: int = 32
    num_attention_heads: int = 32
    rms_norm_eps: float = 1e-5
    rope_theta: float = 10000.0
    max_position_embeddings: int = 4096
    # MoE (medplib_moe_llama.py:63-78; stage-IV values scripts/train_stage4.sh:38-44)
    moe_enable: bool = True
    num_experts: int = 2
    top_k_experts: int = 1
    use_residual: bool = False        # DeepSpeed residual MoE: out * c0 + mlp(x) * c1 (train_ds_medplib.py:131, off in the shipped scripts)
    capacity_factor: float = 1.5
    eval_capacity_factor: float = 2.0
    min_capacity: int = 0
    moe_layers_idx: Optional[List[int]] = None      # None -> all layers ('dense' moe_mode)
    router_aux_loss_coef: float = 0.0
    # DeepSpeed's gate draws (use_rts=True random token selection in top1gating, Gumbel second-expert sampling in top2gating):
    # generated on the device by mp_gate_noise_f32.  Parity tests switch this off or inject the draws into both sides.
    moe_gate_sampling: bool = True
    moe_gate_seed: int = 42
    # CLIP ViT-L/14-336
    clip_image_size: int = 336
    clip_patch_size: int = 14
    clip_hidden_size: int = 1024
    clip_intermediate_size: int = 4096
    clip_num_layers: int = 24
    clip_num_heads: int = 16
    clip_ln_eps: float = 1e-5
    mm_vision_select_layer: int = -2
    mm_use_im_start_end: bool = True
    max_sample_point: int = 512             # region prompts: points sampled per region mask (train_ds_medplib.py:120)
    # ICL front end (medplib_arch.py:67-131; scripts/train_medplib_icl.sh)
    mm_token_compress: bool = False
    mm_compressed_token_count: int = 256
    icl_mask_encoder: bool = False
    mask_encoder_token_count: int = 64
    # SAM-Med2D ViT-B @ 256 (build_sam.py:51-121)
    sam_image_size: int = 256
    sam_embed_dim: int = 768
    sam_depth: int = 12
    sam_num_heads: int = 12
    sam_window: int = 14
    sam_global_attn: tuple = (2, 5, 8, 11)
    sam_out_chans: int = 256
    out_dim: int = 256
    # losses (scripts/train_stage3.sh:24-28)
    ce_loss_weight: float = 1.0
    dice_loss_weight: float = 5.0
    bce_loss_weight: float = 1.0
    iou_loss_weight: float = 0.0
    focal_loss_weight: float = 1.0
    seg_token_idx: int = 32000
    train_mask_decoder: bool = True
    # the mask decoder's upsampler + hypernetwork product as the single fused bf16 kernel (what the reference computes under
    # --precision bf16), training and inference; training differentiates through it with one recomputing backward kernel
    # (autograd_ops.FusedUpsampleMaskFn).  False = six fp32 launches forward, fourteen backward (the strict-parity tail: 2e-4 / 2e-3
    # against the oracle on the same trunk outputs, where the fused form holds 2e-3 / 3e-2).  Needs a 16-multiple token grid.
    fused_bf16_upsampler: bool = True
    # Both RMSNorms of a frozen top-1 MoE decoder layer folded into their consumer GEMMs (round 6): the qkv and the experts'
    # gate|up projections read the raw residual stream, the norm weights are multiplied into the frozen weights' columns once, rstd is applied in the
    # GEMM epilogue.  Moves HF's rounding point (the normalised row is never a bf16 tensor; the folded weight is rounded instead), the MoE gate still
    # sees HF's bf16 h.  Multi-row frozen forwards of 320-row-kernel shapes only; everything else keeps the two norm kernels.
    fold_input_norm: bool = False

    @property
    def head_dim(self):
        return self.hidden_size // self.num_attention_heads

    @property
    def clip_num_patches(self):
        return (self.clip_image_size // self.clip_patch_size) ** 2

    @property
    def image_token_len(self):
        """rows one image placeholder expands to (build_seg_token_mask's default, MedPLIB.py:318-323)."""
        return self.mm_compressed_token_count if self.mm_token_compress else self.clip_num_patches

    @property
    def sam_grid(self):
        return self.sam_image_size // 16

    def moe_layer_set(self):
        if not self.moe_enable:
            return set()
        if self.moe_layers_idx is None:
            return set(range(self.num_hidden_layers))
        return set(self.moe_layers_idx)

    @staticmethod
    def medplib_7b(**kw):
        return MedPLIBConfig(**kw)

    @staticmethod
    def tiny(**kw):
        """Small dims for tests: same structure, every kernel constraint (K % 64, head dims 128 / 64) still exercised."""
        d = dict(vocab_size=515, hidden_size=256, intermediate_size=320, num_hidden_layers=2, num_attention_heads=2,
                 clip_image_size=56, clip_patch_size=14, clip_hidden_size=128, clip_intermediate_size=256, clip_num_layers=3,
                 clip_num_heads=2, seg_token_idx=500, sam_depth=12, moe_gate_sampling=False)
        d.update(kw)
        return MedPLIBConfig(**d)
