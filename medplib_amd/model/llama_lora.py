"""LoRA training of the dense Llama decoder: peft-0.10 adapters on the MLP projections — the targets of
scripts/train_stage3.sh / train_medplib_icl.sh (`--lora_target_modules gate_proj,up_proj,down_proj --lora_r 8 --lora_alpha 16`,
train_ds_medplib.py:262-303) — with the whole decoder backward behind them: the gradient of layer 0's adapters needs the dgrad of
every layer above.

    y = W x + (alpha / r) * B (A dropout(x))        W frozen; A [r, in] (kaiming-uniform), B [out, r] (zeros) trainable

Forward: the frozen projections run on the forward's NT GEMM; each adapter is two thin GEMMs (rank padded to one 64-wide K-tile /
N-tile; gate and up share one pair on the fused, interleaved gate|up matrix).  The forward keeps per layer what the backward reads
(≈ 0.7 GB at 7B, 22 GB for 32 layers — no recomputation).  Backward per layer: dgrad GEMMs on transposed weight copies made once
(`W^T`, +13 GB), `mp_attention_bwd_bf16`, `mp_rmsnorm_bwd_bf16`, `mp_swiglu_pair_bwd_bf16`, RoPE backward = the forward kernel
with −sin, adapter weight gradients by `mp_tn_skinny_f32` (fixed summation order: the step stays bit-reproducible).
autograd sees three Functions: LlamaLoRAFn (the stack), CrossEntropyFn (lm_head + filtered CE), GatherRowsFn (<SEG> rows for the
fp32 tail); everything inside them is this library's kernels.  Targets: any of q/k/v/o/gate/up/down_proj (the shipped scripts' sets).
MoE layers (top-1, one rank): per-expert adapters on the capacity slabs, the routed dgrad, the gate-probability and l_aux gradients
into the gate and a trainable `wg` (scripts/train_stage4.sh's `--sft_modules wg,...`)."""
import math
import os

import numpy as np
import torch

from .. import ops

_SWIGLU_KEEP = os.environ.get("MP_LORA_SWIGLU_KEEP", "1") != "0"     # 0: GEMM + mp_swiglu_pair_fwd_bf16 (A/B runs)
MLP_TARGETS = ("gate_proj", "up_proj", "down_proj")
ALL_TARGETS = ("q_proj", "k_proj", "v_proj", "o_proj") + MLP_TARGETS
# adapter groups: targets that share an input and whose outputs are row blocks of one fused projection get ONE pair of thin GEMMs
GROUPS = {"qkv": ("q_proj", "k_proj", "v_proj"), "o": ("o_proj",), "gu": ("gate_proj", "up_proj"), "down": ("down_proj",)}


# trainable tensors OUTSIDE the decoder stack (their gradients come from other autograd Functions than LlamaLoRAFn)
FRONT_PREFIXES = ("lm_head.weight", "model.embed_tokens.weight", "model.mm_projector.", "model.mm_token_compressor.", "model.region_fea_adapter.",
                  "model.mask_encoder.")


def decoder_param_names(lora):
    """The parameters LlamaLoRAFn takes (and returns gradients for), in `lora.names` order."""
    return [n for n in lora.names if not n.startswith(FRONT_PREFIXES)]


def _module(t):
    return ("self_attn." if t in ("q_proj", "k_proj", "v_proj", "o_proj") else "mlp.") + t


class LoRAState(torch.nn.Module):
    """The adapters of every decoder layer as fp32 nn.Parameters (the engine's flat AdamW buffer adopts them) plus the bf16 padded
    GEMM operands rebuilt from them before each forward."""

    def __init__(self, cfg, llm, r=8, alpha=16, dropout=0.0, targets=MLP_TARGETS, seed=0, train_gate=True, sft_modules=()):
        super().__init__()
        assert r % 8 == 0 and 0 < r <= 16, "lora_r must be 8 or 16 (the shipped scripts' values)"
        assert all(t in ALL_TARGETS for t in targets), f"adapters are built for {ALL_TARGETS}"
        assert not cfg.use_residual or not llm.moe_layers, "training MoE layers: no residual MoE (not yet)"
        assert llm.ep is None or cfg.top_k_experts == 1, "expert-parallel MoE layers under training: top-1 gating"
        self.r, self.alpha, self.p = r, float(alpha), float(dropout)
        self.targets = tuple(t for t in ALL_TARGETS if t in targets)
        self.scaling = self.alpha / r
        self.moe_layers, self.E, self.train_gate = set(llm.moe_layers), cfg.num_experts, bool(train_gate) and bool(llm.moe_layers)
        d, ff, dev = cfg.hidden_size, cfg.intermediate_size, llm.device
        g = torch.Generator().manual_seed(seed)
        self.names, plist = [], []
        for i in range(cfg.num_hidden_layers):
            for t in self.targets:
                fin, fout = (ff, d) if t == "down_proj" else ((d, ff) if t in ("gate_proj", "up_proj") else (d, d))
                bound = 1.0 / math.sqrt(fin)                      # kaiming_uniform_(a=sqrt(5)) on [r, in]
                # peft wraps every Linear whose name matches: in a MoE layer the MLP targets are the experts' projections
                for mod in self._modules_of(i, t):
                    a = (torch.rand(r, fin, generator=g) * 2 - 1) * bound
                    self.names += [f"model.layers.{i}.{mod}.lora_A.default.weight", f"model.layers.{i}.{mod}.lora_B.default.weight"]
                    plist += [torch.nn.Parameter(a.to(dev)), torch.nn.Parameter(torch.zeros(fout, r, device=dev))]
            if self.train_gate and i in self.moe_layers:          # `wg` in --sft_modules (scripts/train_stage4.sh:33)
                self.names.append(f"model.layers.{i}.mlp.deepspeed_moe.gate.wg.weight")
                plist.append(torch.nn.Parameter(llm.layers[i]["wg"].detach().clone()))
        self.norm_names = {}
        for key, short in (("input_layernorm", "ln1"), ("post_attention_layernorm", "ln2")):   # train_stage2.sh's --sft_modules
            if key in sft_modules:
                for i in range(cfg.num_hidden_layers):
                    self.norm_names[(i, short)] = f"model.layers.{i}.{key}.weight"
                    self.names.append(self.norm_names[(i, short)])
                    plist.append(torch.nn.Parameter(llm.layers[i][short].detach().clone()))
        for full in ("lm_head", "embed_tokens"):                  # whole-matrix fine-tuning of --sft_modules (train_stage4.sh:33)
            if full in sft_modules:
                self.names.append("lm_head.weight" if full == "lm_head" else "model.embed_tokens.weight")
                plist.append(torch.nn.Parameter(getattr(llm, full).detach().float().clone()))
        self.params = torch.nn.ParameterList(plist)
        self.index = {n: k for k, n in enumerate(self.names)}
        c = torch.arange(ff, device=dev)
        gate_rows = (c // 32) * 64 + c % 32                       # interleaved row of gate channel c in the fused gate|up matrix
        ar = torch.arange(d, device=dev)
        # rows of each target inside its group's fused output, and the fused output width
        self.rows = {"q_proj": ar, "k_proj": ar + d, "v_proj": ar + 2 * d, "o_proj": ar, "gate_proj": gate_rows, "up_proj": gate_rows + 32,
                     "down_proj": ar}
        self.width = {"qkv": 3 * d, "o": d, "gu": 2 * ff, "down": d}
        self.step = 0
        # set by the engine for data-parallel runs: callable(layer, {name: gradient}) invoked by backward() as soon as a layer's
        # gradients are complete, so their all-reduce overlaps the dgrad of the layers below (engine.Engine._sink)
        self.grad_sink = None
        self._bufs = {}
        self.ext = {}                                           # (layer, group) -> extended weight [out, in + 64] (enable_lora)
        self.p_active = self.p                                  # dropout in effect: p while training, 0 in eval (set per forward)
        self.keep_bits = {}                                     # seed -> lora_dropout mask bytes the forward left for the same step's backward
        self.wgrad_stream = None                                # MP_LORA_WGRAD_STREAM=1: where dA^T = drop(x)^T dt and the gradient unpack run (off the dgrad chain)

    def _modules_of(self, i, t):
        if i in self.moe_layers and t in MLP_TARGETS:
            return [f"mlp.deepspeed_moe.experts.deepspeed_experts.{e}.{t}" for e in range(self.E)]
        return [_module(t)]

    def get(self, i, t, which, e=None):
        mod = self._modules_of(i, t)[e if e is not None else 0]
        return self.params[self.index[f"model.layers.{i}.{mod}.lora_{which}.default.weight"]]

    def add_projector(self, tower):
        """`mm_projector` in --sft_modules (train_stage2.sh): the two Linears train whole; fp32 masters here, bf16 working copies
        (and W2^T for the dgrad) in the tower."""
        self.tower = tower
        pr = tower.proj
        for n, t in (("model.mm_projector.0.weight", pr["w0"]), ("model.mm_projector.0.bias", pr["b0"]),
                     ("model.mm_projector.2.weight", pr["w2"]), ("model.mm_projector.2.bias", pr["b2"])):
            self.index[n] = len(self.names)
            self.names.append(n)
            self.params.append(torch.nn.Parameter(t.detach().float().clone()))
        pr["w2_T"] = pr["w2"].t().contiguous()

    def add_region_adapter(self, tower):
        """`region_fea_adapter` in --sft_modules (scripts/train_stage4.sh:33): the Linear on the raw tower features trains whole."""
        self.region_tower = tower
        for n, t in (("model.region_fea_adapter.weight", tower.region_adapter["w"]), ("model.region_fea_adapter.bias", tower.region_adapter["b"])):
            self.index[n] = len(self.names)
            self.names.append(n)
            self.params.append(torch.nn.Parameter(t.detach().float().clone()))

    def add_mask_encoder(self, enc):
        """`mask_encoder` in --sft_modules (scripts/train_medplib_icl.sh:12): the four convolutions, the projection and the LayerNorm of
        MaskTokenEncoder train.  fp32 masters in this build's own weight layouts (conv weights [cout, 9*cin], tap-major)."""
        self.menc = enc
        items = [("model.mask_encoder.encoder.0.weight", enc.w0), ("model.mask_encoder.encoder.0.bias", enc.b0)]
        for j, i in enumerate((2, 4, 6)):
            items += [(f"model.mask_encoder.encoder.{i}.weight", enc.convs[j][0]), (f"model.mask_encoder.encoder.{i}.bias", enc.convs[j][1])]
        items += [("model.mask_encoder.proj.weight", enc.proj_w), ("model.mask_encoder.proj.bias", enc.proj_b),
                  ("model.mask_encoder.norm.weight", enc.norm[0]), ("model.mask_encoder.norm.bias", enc.norm[1])]
        for n, t in items:
            self.index[n] = len(self.names)
            self.names.append(n)
            self.params.append(torch.nn.Parameter(t.detach().float().clone()))

    def add_token_compressor(self, comp):
        """`mm_token_compressor` in --sft_modules (scripts/train_medplib_icl.sh:8): LayerNorm + Linear train; fp32 masters here."""
        self.comp = comp
        for n, t in (("model.mm_token_compressor.norm.weight", comp.norm[0]), ("model.mm_token_compressor.norm.bias", comp.norm[1]),
                     ("model.mm_token_compressor.proj.weight", comp.proj_w), ("model.mm_token_compressor.proj.bias", comp.proj_b)):
            self.index[n] = len(self.names)
            self.names.append(n)
            self.params.append(torch.nn.Parameter(t.detach().float().clone()))
        comp.proj_w_T = comp.proj_w.t().contiguous()

    def full_param(self, name):
        k = self.index.get(name)
        return None if k is None else self.params[k]

    def sync_model(self, llm):
        """bf16 working copies of the fully fine-tuned matrices (and lm_head^T for its dgrad) from their fp32 masters; once per step.
        Trainable norm weights are fp32 in the model already: its tensors are re-pointed to the parameters."""
        for (i, short), n in self.norm_names.items():
            llm.layers[i][short] = self.params[self.index[n]].data
        if self.train_gate:                                   # a trained gate: merge / export run without a training forward
            for i in self.moe_layers:
                self.gate_weight(i, llm)
        p = self.full_param("lm_head.weight")
        if p is not None:
            llm.lm_head.copy_(p.detach())
            llm.lm_head_T[:, :p.shape[0]] = llm.lm_head.t()
        p = self.full_param("model.embed_tokens.weight")
        if p is not None:
            llm.embed_tokens.copy_(p.detach())
        if self.full_param("model.mask_encoder.proj.weight") is not None:
            e, fp = self.menc, self.full_param
            e.w0, e.b0 = fp("model.mask_encoder.encoder.0.weight").data, fp("model.mask_encoder.encoder.0.bias").data
            for j, i in enumerate((2, 4, 6)):
                e.convs[j][0].copy_(fp(f"model.mask_encoder.encoder.{i}.weight").detach())
                e.convs[j] = (e.convs[j][0], fp(f"model.mask_encoder.encoder.{i}.bias").data)
            e.proj_w.copy_(fp("model.mask_encoder.proj.weight").detach())
            e.proj_b = fp("model.mask_encoder.proj.bias").data
            e.norm = (fp("model.mask_encoder.norm.weight").data, fp("model.mask_encoder.norm.bias").data)
        if self.full_param("model.region_fea_adapter.weight") is not None:
            ra = self.region_tower.region_adapter
            ra["w"].copy_(self.full_param("model.region_fea_adapter.weight").detach())
            ra["b"] = self.full_param("model.region_fea_adapter.bias").data
        if self.full_param("model.mm_token_compressor.proj.weight") is not None:
            c = self.comp
            c.proj_w.copy_(self.full_param("model.mm_token_compressor.proj.weight").detach())
            c.proj_w_T.copy_(c.proj_w.t())
            c.proj_b = self.full_param("model.mm_token_compressor.proj.bias").data
            c.norm = (self.full_param("model.mm_token_compressor.norm.weight").data, self.full_param("model.mm_token_compressor.norm.bias").data)
        if self.full_param("model.mm_projector.0.weight") is not None:
            pr = self.tower.proj
            pr["w0"].copy_(self.full_param("model.mm_projector.0.weight").detach())
            pr["w2"].copy_(self.full_param("model.mm_projector.2.weight").detach())
            pr["b0"] = self.full_param("model.mm_projector.0.bias").data
            pr["b2"] = self.full_param("model.mm_projector.2.bias").data
            pr["w2_T"].copy_(pr["w2"].t())

    def gate_weight(self, i, llm):
        """The gate of MoE layer i: the trainable fp32 copy when `wg` trains (the model's tensor is re-pointed to it), else the model's."""
        k = self.index.get(f"model.layers.{i}.mlp.deepspeed_moe.gate.wg.weight")
        if k is not None:
            llm.layers[i]["wg"] = self.params[k].data
        return llm.layers[i]["wg"]

    def peft_state_dict(self):
        """The adapters under peft's key names (`base_model.model.<module>.lora_{A,B}.default.weight`), bf16 like a bf16 peft model saves
        them; `lora.merge_lora_state_dict` folds such a dict into the base weights (merge_lora_weights_and_save_hf_model_moe.py:270-345)."""
        return {"base_model.model." + n: p.detach().to(torch.bfloat16) for n, p in zip(self.names, self.params)}

    def load_peft_state_dict(self, sd):
        for n, p in zip(self.names, self.params):
            k = n if n in sd else "base_model.model." + n
            p.data.copy_(sd[k].to(p.dtype))

    @torch.no_grad()
    def merge_into(self, llm):
        """peft merge_and_unload (merge_lora_weights_and_save_hf_model_moe.py:339): W += scaling * B A for every adapter, into the
        model's bf16 weights in place (the fused / interleaved / per-expert row layouts included); fully fine-tuned matrices are synced."""
        self.sync_model(llm)
        for i, lw in enumerate(llm.layers):
            for t in self.targets:
                grp = next(g for g, mem in GROUPS.items() if t in mem)
                local = llm.ep.local_expert_ids() if (llm.ep is not None and lw[grp].dim() == 3) else None
                for e, _ in enumerate(self._modules_of(i, t)):
                    if local is not None and e not in local:
                        continue                                  # expert parallelism: this rank holds (and merges) its own experts only
                    delta = self.scaling * (self.get(i, t, "B", e).float() @ self.get(i, t, "A", e).float())
                    W = lw[grp] if lw[grp].dim() == 2 else lw[grp][e - (local[0] if local is not None else 0)]
                    rows = self.rows[t]
                    W[rows] = (W[rows].float() + delta).to(W.dtype)
            for k in ("qkv", "o", "gu", "down"):
                if k + "_T" in lw:
                    lw[k + "_T"] = _transposed(lw[k])
                if k + "_x" in lw:
                    lw[k + "_x"][..., :lw[k].shape[-1]].copy_(lw[k])
        llm.refresh_fused_qkv()                                   # the RoPE-interleaved copies follow the merged q / k / v weights

    def padded(self, i):
        """bf16 GEMM operands of layer i per adapter group: (A [64, in], A^T [in, 64], B [out, 64], B^T [64, out], R, targets) — with a
        leading expert axis for the MLP groups of a MoE layer, and B stored * scaling there (the batched GEMM has no alpha).  The
        buffers persist (zeroed once: the padding never changes); each step one pack kernel per adapter rewrites its slices.
        R = rank of the fused pair (targets x r), rounded up to what the wgrad kernel takes."""
        r, dev, bf = self.r, self.rows["q_proj"].device, torch.bfloat16
        moe = i in self.moe_layers
        if i not in self._bufs:
            self._bufs[i] = {}
            for grp, members in GROUPS.items():
                tg = [t for t in members if t in self.targets]
                if tg:
                    fin, W = self.get(i, tg[0], "A").shape[1], self.width[grp]
                    lead = (self.E,) if (moe and grp in ("gu", "down")) else ()
                    self._bufs[i][grp] = (torch.zeros(lead + (64, fin), dtype=bf, device=dev), torch.zeros(lead + (fin, 64), dtype=bf, device=dev),
                                          torch.zeros(lead + (W, 64), dtype=bf, device=dev), torch.zeros(lead + (64, W), dtype=bf, device=dev), tg)
        out = {}
        # pack_all() already rewrote every slice this step AND nothing has written the parameters since (the optimizer step bumps ops.PARAM_EPOCH;
        # a caller between optimizer.step() and the next forward — a merge, an eval helper — must not get the previous values' images)
        fresh = getattr(self, "_packed_at", None) == self.step and self.step > 0 and getattr(self, "_packed_epoch", None) == ops.PARAM_EPOCH
        for grp, (A, AT, B, BT, tg) in self._bufs[i].items():
            batched = A.dim() == 3
            for k, t in enumerate(tg):
                for e in range(self.E if batched else 1):
                    if fresh:
                        continue
                    a, b = self.get(i, t, "A", e).detach(), self.get(i, t, "B", e).detach()
                    if batched:
                        wx = self.ext.get((i, grp))                # [E, out, in + 64]: expert e's scaling * B behind its frozen weight
                        ops.lora_pack(a, b, self.rows[t], A[e], AT[e], B[e], BT[e], k * r, bscale=self.scaling,
                                      Bx=None if wx is None else wx[e][:, wx.shape[2] - 64:], xscale=self.scaling)
                    else:
                        wx = self.ext.get((i, grp))                # dense group: scaling * B also lands in [W | scaling B]'s last 64 columns
                        ops.lora_pack(a, b, self.rows[t], A, AT, B, BT, k * r, Bx=None if wx is None else wx[:, wx.shape[1] - 64:],
                                      xscale=self.scaling)
            R = len(tg) * r
            out[grp] = (A, AT, B, BT, 8 if R <= 8 else 16 if R <= 16 else 32 if R <= 32 else 64, tg)
        return out


_PACK_DESC = np.dtype([("a", "<u8"), ("b", "<u8"), ("rows", "<u8"), ("A", "<u8"), ("AT", "<u8"), ("B", "<u8"), ("BT", "<u8"), ("Bx", "<u8"), ("ldbx", "<i8"),
                       ("r", "<i4"), ("fin", "<i4"), ("fout", "<i4"), ("k0", "<i4"), ("W", "<i4"), ("bscale", "<f4"), ("xscale", "<f4"), ("pad", "<i4")])
assert _PACK_DESC.itemsize == 104


def pack_all(lora, n_layers):
    """Every adapter's slices of the padded bf16 operands rewritten in ONE launch (mp_lora_pack_batched) instead of one mp_lora_pack per adapter and
    layer (96 per step at the shipped stage-III configuration).  The device-side descriptor table is rebuilt only when a pointer moved
    (the engine re-homes the parameters into its flat buffer once).  Same kernel body, same values."""
    for i in range(n_layers):                                  # make sure every layer's persistent buffers exist (first call)
        if i not in lora._bufs:
            lora.padded(i)
    recs = []
    r = lora.r
    for i in range(n_layers):
        moe = i in lora.moe_layers
        for grp, (A, AT, B, BT, tg) in lora._bufs[i].items():
            batched = A.dim() == 3
            wx = lora.ext.get((i, grp))
            for k, t in enumerate(tg):
                for e in range(lora.E if batched else 1):
                    a, b = lora.get(i, t, "A", e).detach(), lora.get(i, t, "B", e).detach()
                    Ae, ATe, Be, BTe = (A[e], AT[e], B[e], BT[e]) if batched else (A, AT, B, BT)
                    bx = None if wx is None else (wx[e][:, wx.shape[2] - 64:] if batched else wx[:, wx.shape[1] - 64:])
                    recs.append((a.data_ptr(), b.data_ptr(), lora.rows[t].data_ptr(), Ae.data_ptr(), ATe.data_ptr(), Be.data_ptr(), BTe.data_ptr(),
                                 0 if bx is None else bx.data_ptr(), 0 if bx is None else bx.stride(0), a.shape[0], a.shape[1], b.shape[0], k * r, Be.shape[0],
                                 lora.scaling if batched else 1.0, lora.scaling, 0))
    key = tuple(recs)
    if getattr(lora, "_pack_key", None) != key:
        arr = np.array(recs, dtype=_PACK_DESC)
        lora._pack_tab = torch.from_numpy(arr.view(np.uint8).reshape(-1).copy()).to(lora.rows["q_proj"].device)
        lora._pack_key, lora._pack_max = key, max(int(x[9]) * int(x[10]) + int(x[11]) * int(x[9]) for x in recs)
    ops.lib().call("mp_lora_pack_batched", lora._pack_tab.data_ptr(), len(recs), int(lora._pack_max), ops._stream())
    lora._packed_at, lora._packed_epoch = lora.step, ops.PARAM_EPOCH


def _transposed(w):
    """The transposed copy of a projection weight the dgrad GEMMs read as their W operand; 2-D copies get a row stride off multiples of
    8 KiB (ops.padded_rows: qkv^T [d, 3d] has 24 KiB rows)."""
    if w.dim() == 2:
        out = ops.padded_rows(w.shape[1], w.shape[0], w.device, w.dtype)
        out.copy_(w.t())
        return out
    return w.transpose(-1, -2).contiguous()


def _extended(w):
    """[out, in] -> [out, in + 64] with the weight in the first `in` columns and zeros behind (filled with scaling * B per step)."""
    out = torch.zeros(w.shape[:-1] + (w.shape[-1] + 64,), dtype=w.dtype, device=w.device)      # experts: [E, out, in + 64]
    out[..., :w.shape[-1]].copy_(w)
    return out


def enable_lora(llm, cfg, r=8, alpha=16, dropout=0.0, targets=MLP_TARGETS, seed=0, train_gate=True, sft_modules=()):
    """Attach adapters to a LlamaStack and make the transposed weight copies the dgrad GEMMs read."""
    llm.lora = LoRAState(cfg, llm, r, alpha, dropout, targets, seed, train_gate, tuple(sft_modules))
    for i, lw in enumerate(llm.layers):
        for k in ("qkv", "o", "gu", "down"):
            lw[k + "_T"] = _transposed(lw[k])                            # experts: [E, out, in] -> [E, in, out]
            if (lw[k].dim() == 2 or llm.ep is None) and any(t in llm.lora.targets for t in GROUPS[k]):
                # the adapter's up-projection as a K-extension of the frozen weight: [W | scaling B] (ops.lora_down writes the matching
                # 64 columns of the activations), so base + adapter is ONE GEMM over K + 64
                lw[k + "_x"] = _extended(lw[k])
                llm.lora.ext[(i, k)] = lw[k + "_x"]
    V, d = llm.lm_head.shape
    vp = (V + 63) // 64 * 64
    llm.lm_head_T = torch.zeros(d, vp, dtype=torch.bfloat16, device=llm.device)
    llm.lm_head_T[:, :V] = llm.lm_head.t()
    llm.sin_neg = (-llm.sin).contiguous()
    return llm.lora


def _ext_rows(T, K, dev):
    """The row-padded input of an extended projection: (whole [T, K + 64] buffer, its [T, K] activation part, its [T, 64] adapter part)."""
    buf = torch.empty((T, K + 64), dtype=torch.bfloat16, device=dev)
    return buf, buf[:, :K], buf[:, K:]


_ROPE_IN_ATTN_BWD = os.environ.get("MP_ATTN_BWD_ROPE", "1") != "0"      # A/B: 0 = mp_rope_qk_bf16 over the fused gradient after the attention backward
_FUSE_SWSK = os.environ.get("MP_LORA_FUSE_SWIGLU_SKINNY", "1") != "0"   # A/B: 0 = d gate|up written by one kernel, read back by the gate|up adapter's products
_FUSE_NORM_UP = os.environ.get("MP_LORA_FUSE_NORM_UP", "1") != "0"      # A/B: 0 = mp_lora_up_add_bf16, then mp_rmsnorm_bwd_bf16
_FUSE_DY = os.environ.get("MP_LORA_FUSE_DY", "1") != "0"                # A/B: 0 = dy B and dy^T t as two kernels, two reads of dy
_KEEP_BITS = os.environ.get("MP_LORA_KEEP_BITS", "1") != "0"            # A/B: 0 = every kernel regenerates the lora_dropout mask from the seed
_PRUNE_ROWS = os.environ.get("MP_PRUNE_LAST_MLP", "1") != "0"            # A/B: 0 = the last layer's MLP on every row
_UNPACK_PARTIALS = os.environ.get("MP_LORA_UNPACK_PARTIALS", "1") != "0"  # A/B: 0 = a reduce launch per weight-gradient product, then the unpack
_PACK_BATCHED = os.environ.get("MP_LORA_PACK_BATCHED", "1") != "0"       # A/B: 0 = one mp_lora_pack launch per adapter and layer
_WGRAD_STREAM = os.environ.get("MP_LORA_WGRAD_STREAM", "0") == "1"      # A/B: 1 = the adapters' weight-gradient products and their unpack on a side stream
_FUSE_UP_SWIGLU = os.environ.get("MP_FUSE_UP_SWIGLU", "1") != "0"      # A/B: 0 = lora_up_add then swiglu_pair_bwd (two passes over d_act)


def _adapter_down(lora, ops_pad, x, t, seed):
    """t = bf16(dropout(x) A^T) into the extension columns -> x (the wgrad regenerates the mask from the seed: nothing dropped is stored)."""
    A, _, _, _, R, _ = ops_pad
    kb = None
    if lora.p_active > 0 and _KEEP_BITS and x.shape[1] % 256 == 0:
        # the mask as bytes (1 / 16 of x): the two backward kernels that need the same mask read it instead of hashing again
        kb = lora.keep_bits[seed] = ops.keep_bits_for(x)
    ops.lora_down(x, A, t, R, lora.p_active, seed, keep_bits=kb)
    return x


def _zeros(shape, dev):
    return torch.zeros(shape, dtype=torch.bfloat16, device=dev)


def _adapter_fwd_moe(lora, ops_pad, xbuf, ybuf, kept, seed):
    """Per-expert adapters on the capacity slabs: ybuf + (dropout(xbuf) A_e^T) (scaling B_e)^T -> (y', x_dropped, t)."""
    A, _, B, _, _, _ = ops_pad
    E, cap, _ = xbuf.shape
    xd = ops.dropout_bf16(xbuf, lora.p_active, seed) if lora.p_active > 0 else xbuf
    t = ops.gemm_batched(xd, A, _zeros((E, cap, 64), xbuf.device), m_dev=kept)
    return ops.gemm_batched_res(t, B, ybuf, _zeros(ybuf.shape, xbuf.device), m_dev=kept), xd, t


def _moe_fwd(llm, lora, i, lw, pad, h2, x_mid, s, seed):
    """DeepSpeed MoE layer (top-1) in training-with-adapters mode on capacity slabs that are zero where no token sits, so every
    row of every slab is finite and rows without a token contribute nothing to the weight gradients."""
    cfg = llm.cfg
    T, d = h2.shape
    E, ff = cfg.num_experts, cfg.intermediate_size
    cap = llm.capacity(T)
    wg = lora.gate_weight(i, llm)
    k = cfg.top_k_experts
    logits, gates = ops.moe_gate(h2, wg)
    if k == 1:
        expert, slot, weight, kept, counts, l_aux = ops.moe_route_top1(gates, cap, llm._gate_draws(i, T, E, gumbel=False))
    else:
        expert, slot, weight, kept, _, l_aux = ops.moe_route_top2(gates, logits, cap, llm._gate_draws(i, T, E, gumbel=True))
        counts = torch.bincount(expert[:T].long(), minlength=E)    # l_aux is built on the FIRST choices (top2gating's mask1)
    dev = h2.device
    empty = lambda *shape: torch.empty(shape, dtype=torch.bfloat16, device=dev)
    if "gu_x" in lw:
        # fused adapter branch on the capacity slabs (as the dense layers, per expert): nothing is zero-filled -- rows beyond an expert's
        # count hold whatever the allocator left, every reduction over rows is limited by the device-side count (GEMMs: m_dev, weight
        # gradients: rows_dev) and the combine only reads routed slots
        bufx = empty(E, cap, d + 64)
        buf = ops.moe_dispatch(h2, expert, slot, E, cap, buf=bufx[:, :, :d], top_k=k)
        for e in range(E):
            ops.lora_down(buf[e], pad["gu"][0][e], bufx[e][:, d:], pad["gu"][4], lora.p_active, seed * 16 + e, rows_dev=kept[e:e + 1])
        gu = ops.gemm_batched(bufx, lw["gu_x"], empty(E, cap, 2 * ff), m_dev=kept)
        s["bufd"], s["t_gu"] = buf, bufx[:, :, d:]
    else:
        buf = ops.moe_dispatch(h2, expert, slot, E, cap, buf=_zeros((E, cap, d), dev), top_k=k)
        gu = ops.gemm_batched(buf, lw["gu"], _zeros((E, cap, 2 * ff), dev), m_dev=kept)
        if "gu" in pad:
            gu, s["bufd"], s["t_gu"] = _adapter_fwd_moe(lora, pad["gu"], buf, gu, kept, seed)
    if "down_x" in lw:
        actx = empty(E, cap, ff + 64)
        act = actx[:, :, :ff]
        ops.swiglu_pair_fwd(gu.view(E * cap, 2 * ff), out=actx.view(E * cap, ff + 64)[:, :ff], counts=kept, cap=cap)
        for e in range(E):
            ops.lora_down(act[e], pad["down"][0][e], actx[e][:, ff:], pad["down"][4], lora.p_active, (seed + 1) * 16 + e, rows_dev=kept[e:e + 1])
        y = ops.gemm_batched(actx, lw["down_x"], empty(E, cap, d), m_dev=kept)
        s["actd"], s["t_d"] = act, actx[:, :, ff:]
    else:
        act = ops.swiglu_pair_fwd(gu.view(E * cap, 2 * ff)).view(E, cap, ff)
        y = ops.gemm_batched(act, lw["down"], _zeros((E, cap, d), dev), m_dev=kept)
        if "down" in pad:
            y, s["actd"], s["t_d"] = _adapter_fwd_moe(lora, pad["down"], act, y, kept, seed + 1)
    s["fused_moe"] = ("gu_x" in lw, "down_x" in lw)
    s.update(moe=True, h2=h2, gates=gates, expert=expert, slot=slot, weight=weight, kept=kept, counts=counts, gu=gu, y=y, cap=cap, wg=wg)
    return ops.moe_combine(y, expert, slot, weight, x_mid, cap, top_k=k), l_aux


def _ep_pad(ops_pad, eg, ep):
    """The padded adapter operands of ONE global expert `eg`, broadcast (stride 0) over the `ep` source-rank slabs it serves."""
    A, AT, B, BT, R, tg = ops_pad
    x = lambda t: t[eg].unsqueeze(0).expand(ep, -1, -1)
    return (x(A), x(AT), x(B), x(BT), R, tg)


def _moe_fwd_ep(llm, lora, i, lw, pad, h2, x_mid, s, seed):
    """_moe_fwd with the experts SHARDED over an expert-parallel group (DeepSpeed `ep_size` > 1, medplib_moe_llama.py:604-614; top-1):
    this rank routes its own tokens, the routed rows travel to the ranks that own their experts (expert_parallel.py: one all-to-all,
    row counts in the slab headers), every local expert runs over the `ep` slabs it received — base projections as one batched GEMM
    per expert with the expert's matrix broadcast over the slabs, its LoRA adapters the same way — and the outputs travel back.
    Slabs are zero where no token sits, so every row is finite and empty rows add nothing to the weight gradients."""
    cfg, ep = llm.cfg, llm.ep
    if cfg.top_k_experts != 1:
        # checked HERE, where the routing happens: enable_lora() and enable_expert_parallel() may be called in either order, and the
        # frozen / eval path (`LlamaStack._mlp`) does route top-2 — training top-1 next to it would be a silently different model
        raise NotImplementedError(f"expert-parallel MoE layers under LoRA training route top-1 only; top_k_experts = {cfg.top_k_experts} "
                                  "(the reference driver's default) needs ep_size 1, or train with --top_k_experts 1")
    T, d = h2.shape
    E, ff = cfg.num_experts, cfg.intermediate_size
    cap = llm.capacity(T)
    wg = lora.gate_weight(i, llm)
    logits, gates = ops.moe_gate(h2, wg)
    expert, slot, weight, kept, counts, l_aux = ops.moe_route_top1(gates, cap, llm._gate_draws(i, T, E, gumbel=False))
    capx = ep.exchange_capacity(cap, key=llm.gate_pass)
    buf = ops.moe_dispatch(h2, expert, slot, E, capx + 1, buf=_zeros((E, capx + 1, d), h2.device))
    recv, rcounts = ep.dispatch(buf, kept)                                   # [ep, El, capx + 1, d], [ep, El]
    xin = recv[:, :, :capx].permute(1, 0, 2, 3).contiguous()                 # [El, ep, capx, d]: local expert e's slabs are xin[e]
    rc = rcounts.t().contiguous()                                            # [El, ep]
    ids = ep.local_expert_ids()
    y_loc = torch.empty((ep.E_local, ep.ep, capx, d), dtype=torch.bfloat16, device=h2.device)
    loc = []
    for e, eg in enumerate(ids):
        w_gu = lw["gu"][e].unsqueeze(0).expand(ep.ep, -1, -1)               # lw holds this rank's E_local experts only
        w_dn = lw["down"][e].unsqueeze(0).expand(ep.ep, -1, -1)
        st = {}
        gu = ops.gemm_batched(xin[e], w_gu, _zeros((ep.ep, capx, 2 * ff), h2.device), m_dev=rc[e])
        if "gu" in pad:
            gu, st["bufd"], st["t_gu"] = _adapter_fwd_moe(lora, _ep_pad(pad["gu"], eg, ep.ep), xin[e], gu, rc[e], seed * 16 + eg)
        act = ops.swiglu_pair_fwd(gu.view(ep.ep * capx, 2 * ff)).view(ep.ep, capx, ff)
        y = ops.gemm_batched(act, w_dn, _zeros((ep.ep, capx, d), h2.device), m_dev=rc[e])
        if "down" in pad:
            y, st["actd"], st["t_d"] = _adapter_fwd_moe(lora, _ep_pad(pad["down"], eg, ep.ep), act, y, rc[e], (seed + 1) * 16 + eg)
        y_loc[e] = y
        st.update(gu=gu, x=xin[e])
        loc.append(st)
    y_all = ep.combine(y_loc.permute(1, 0, 2, 3).contiguous())              # [E, capx, d]: every global expert's rows for MY tokens
    s.update(moe=True, ep=True, h2=h2, gates=gates, expert=expert, slot=slot, weight=weight, kept=kept, counts=counts, y=y_all, cap=capx,
             wg=wg, loc=loc, rc=rc)
    return ops.moe_combine(y_all, expert, slot, weight, x_mid, capx, top_k=1), l_aux


def _moe_bwd_ep(llm, lora, i, lw, s, dx, d_aux, grads, take_e):
    """Backward of _moe_fwd_ep: the output-row gradients travel to the experts' ranks (`exchange`), each local expert's dgrad and
    adapter gradients run over its slabs (the adapter gradients of the `ep` slabs are summed: they are ONE expert's parameters), the
    input-row gradients travel back (`combine`), then the gate as in _moe_bwd.  Gradients of non-local experts' adapters do not
    exist on this rank (the engine's SUM all-reduce collects each expert's from its owners)."""
    cfg, ep = llm.cfg, llm.ep
    E, ff, capx = cfg.num_experts, cfg.intermediate_size, s["cap"]
    T, d = dx.shape
    pad, rc = s["pad"], s["rc"]
    d_y, d_w = ops.moe_combine_bwd(dx, s["y"], s["expert"], s["slot"], s["weight"], capx, top_k=1)        # [E, capx, d]
    dy_loc = ep.exchange(d_y).permute(1, 0, 2, 3).contiguous()                                             # [El, ep, capx, d]
    dbuf_loc = torch.empty_like(dy_loc)
    for e, eg in enumerate(ep.local_expert_ids()):
        st = s["loc"][e]
        dn_T = lw["down_T"][e].unsqueeze(0).expand(ep.ep, -1, -1)
        gu_T = lw["gu_T"][e].unsqueeze(0).expand(ep.ep, -1, -1)
        d_act = ops.gemm_batched(dy_loc[e], dn_T, _zeros((ep.ep, capx, ff), dx.device), m_dev=rc[e])
        if "down" in pad:
            d_act, dB, dAT = _adapter_bwd_moe(lora, _ep_pad(pad["down"], eg, ep.ep), dy_loc[e], st["actd"], st["t_d"], d_act, rc[e],
                                              (s["seed"] + 1) * 16 + eg)
            take_e(i, pad["down"], {eg: sum(dB[1:], dB[0])}, {eg: sum(dAT[1:], dAT[0])})
        d_gu = ops.swiglu_pair_bwd(st["gu"].view(ep.ep * capx, 2 * ff), d_act.view(ep.ep * capx, ff)).view(ep.ep, capx, 2 * ff)
        d_in = ops.gemm_batched(d_gu, gu_T, _zeros((ep.ep, capx, d), dx.device), m_dev=rc[e])
        if "gu" in pad:
            d_in, dB, dAT = _adapter_bwd_moe(lora, _ep_pad(pad["gu"], eg, ep.ep), d_gu, st["bufd"], st["t_gu"], d_in, rc[e], s["seed"] * 16 + eg)
            take_e(i, pad["gu"], {eg: sum(dB[1:], dB[0])}, {eg: sum(dAT[1:], dAT[0])})
        dbuf_loc[e] = d_in
    d_buf = ep.combine(dbuf_loc.permute(1, 0, 2, 3).contiguous())                                          # [E, capx, d]
    ones = torch.ones(T, dtype=torch.float32, device=dx.device)
    d_h2 = ops.moe_combine(d_buf, s["expert"], s["slot"], ones, None, capx, top_k=1)
    dl = ops.moe_gate_bwd(s["gates"], s["expert"], s["slot"], d_w, s["counts"], d_aux, 1.0, top_k=1)
    ops.moe_gate_dgrad_(dl, s["wg"], d_h2)
    name = f"model.layers.{i}.mlp.deepspeed_moe.gate.wg.weight"
    if name in lora.index:
        dlb = torch.zeros((T, 8), dtype=torch.bfloat16, device=dx.device)
        dlb[:, :E] = dl
        grads[name] = ops.tn_skinny(s["h2"], dlb, 8, 1.0)[:, :E].t()
    return d_h2


def _adapter_bwd_moe(lora, ops_pad, dy, xd, t, dx, kept, seed):
    """Per-expert adapter gradients on the slabs: (dx', [dB_e [out, R]], [dA_e^T [in, R]])."""
    A, AT, B, BT, R, _ = ops_pad
    E, cap, _ = dy.shape
    dt = ops.gemm_batched(dy, BT, _zeros((E, cap, 64), dy.device), m_dev=kept)            # scaling rides in the packed B
    dB = [ops.tn_skinny(dy[e], t[e], R, lora.scaling) for e in range(E)]
    dAT = [ops.tn_skinny(xd[e], dt[e], R, 1.0) for e in range(E)]
    if lora.p_active > 0:
        dxa = ops.dropout_bf16(ops.gemm_batched(dt, AT, _zeros(dx.shape, dy.device), m_dev=kept), lora.p_active, seed)
        return ops.add3(dx, dxa), dB, dAT
    return ops.gemm_batched_res(dt, AT, dx, _zeros(dx.shape, dy.device), m_dev=kept), dB, dAT


def _adapter_bwd_moe_fused(lora, ops_pad, dy, x, t, dx, kept, seed):
    """Per-expert adapter gradients for the fused forward (x = the UNdropped slabs, t = their extension columns): dt = dY (scaling B)
    by the down-projection kernel, weight gradients limited to each expert's routed rows, dx += dropout(dt A) in place."""
    A, AT, B, BT, R, _ = ops_pad
    E, cap, _ = dy.shape
    dB, dAT = [], []
    for e in range(E):
        cnt = kept[e:e + 1]
        dt = ops.lora_down(dy[e], BT[e], torch.empty((cap, 64), dtype=torch.bfloat16, device=dy.device), R, rows_dev=cnt)    # scaling rides in the packed B
        dB.append(ops.tn_skinny(dy[e], t[e], R, lora.scaling, rows_dev=cnt))
        dAT.append(ops.tn_skinny(x[e], dt, R, 1.0, lora.p_active, seed * 16 + e, rows_dev=cnt))
        if R <= 32:
            ops.lora_up_add(dt, AT[e], dx[e], R, lora.p_active, seed * 16 + e, rows_dev=cnt)
        else:
            dxa = ops.gemm(dt, AT[e])
            dx[e].copy_(ops.add3(dx[e].contiguous(), ops.dropout_bf16(dxa, lora.p_active, seed * 16 + e) if lora.p_active > 0 else dxa))
    return dx, dB, dAT


def _moe_bwd(llm, lora, i, lw, s, dx, d_aux, grads, take_e):
    """Backward of _moe_fwd: routed dgrad through the experts (+ their adapters), the combine weights' gradient into the gate
    (softmax probability of the chosen expert) together with l_aux's, the gate's input gradient, and d wg.  -> d_h2 [T, d]."""
    cfg = llm.cfg
    E, ff, cap = cfg.num_experts, cfg.intermediate_size, s["cap"]
    T, d = dx.shape
    pad, kept = s["pad"], s["kept"]
    k = cfg.top_k_experts
    d_y, d_w = ops.moe_combine_bwd(dx, s["y"], s["expert"], s["slot"], s["weight"], cap, top_k=k)
    fused_gu, fused_down = s.get("fused_moe", (False, False))
    slab = (lambda *shape: torch.empty(shape, dtype=torch.bfloat16, device=dx.device)) if fused_gu and fused_down else (lambda *shape: _zeros(shape, dx.device))
    d_act = ops.gemm_batched(d_y, lw["down_T"], slab(E, cap, ff), m_dev=kept)
    if "down" in pad:
        d_act, dB, dAT = (_adapter_bwd_moe_fused if fused_down else _adapter_bwd_moe)(lora, pad["down"], d_y, s["actd"], s["t_d"], d_act, kept, s["seed"] + 1)
        take_e(i, pad["down"], dB, dAT)
    d_gu = ops.swiglu_pair_bwd(s["gu"].view(E * cap, 2 * ff), d_act.view(E * cap, ff), counts=kept if (fused_gu and fused_down) else None,
                               cap=cap).view(E, cap, 2 * ff)
    d_buf = ops.gemm_batched(d_gu, lw["gu_T"], slab(E, cap, d), m_dev=kept)
    if "gu" in pad:
        d_buf, dB, dAT = (_adapter_bwd_moe_fused if fused_gu else _adapter_bwd_moe)(lora, pad["gu"], d_gu, s["bufd"], s["t_gu"], d_buf, kept, s["seed"])
        take_e(i, pad["gu"], dB, dAT)
    ones = torch.ones(T * k, dtype=torch.float32, device=dx.device)
    d_h2 = ops.moe_combine(d_buf, s["expert"], s["slot"], ones, None, cap, top_k=k)        # rows back to their tokens (dropped: 0)
    dl = ops.moe_gate_bwd(s["gates"], s["expert"], s["slot"], d_w, s["counts"], d_aux, 1.0, top_k=k)
    ops.moe_gate_dgrad_(dl, s["wg"], d_h2)
    name = f"model.layers.{i}.mlp.deepspeed_moe.gate.wg.weight"
    if name in lora.index:
        dlb = torch.zeros((T, 8), dtype=torch.bfloat16, device=dx.device)
        dlb[:, :E] = dl
        grads[name] = ops.tn_skinny(s["h2"], dlb, 8, 1.0)[:, :E].t()
    return d_h2


def forward_train(llm, embeds, key_valid):
    """The decoder forward in training-with-adapters mode -> (last_hidden [B,S,d], saved)."""
    cfg, lora = llm.cfg, llm.lora
    B, S, d = embeds.shape
    H, D = cfg.num_attention_heads, cfg.head_dim
    T = B * S
    x = embeds.reshape(T, d)
    lora.step += 1
    lora.p_active = lora.p if llm.training else 0.0
    lora.keep_bits = {}                                         # seed -> the mask bytes _adapter_down left for this step's backward
    if _PACK_BATCHED:
        pack_all(lora, len(llm.layers))
    llm.gate_pass += 1
    saved, aux = [], []
    needed = getattr(llm, "needed_rows", None)                  # (rows int64, mask uint8) of the output rows something reads, or None
    if needed is not None and needed[1].numel() != T:
        needed = None
    for i, lw in enumerate(llm.layers):
        pad = lora.padded(i)
        s = {"x": x, "pad": pad}
        seed = (lora.step * 4096 + i) * 4
        if "qkv_x" in lw:
            h1x, h1, t1 = _ext_rows(T, d, x.device)
            ops.rmsnorm(x, lw["ln1"], cfg.rms_norm_eps, out=h1)
            s["h1d"], s["t_qkv"] = _adapter_down(lora, pad["qkv"], h1, t1, seed + 2), t1
            qkv = ops.gemm(h1x, lw["qkv_x"], out=ops.padded_rows(T, 3 * d, x.device))
            ops.rope_qk_(qkv, llm.cos, llm.sin, S, H, D)
        else:
            h1 = ops.rmsnorm(x, lw["ln1"], cfg.rms_norm_eps)
            if "qkv_rope" in lw:                                    # no adapters on q / k / v: RoPE in the GEMM's epilogue as in the frozen forward
                qkv = ops.gemm_qkv_rope(h1, lw["qkv_rope"], llm.cos, llm.sin, S, H, D, out=ops.padded_rows(T, 3 * d, x.device))
            else:
                qkv = ops.gemm(h1, lw["qkv"], out=ops.padded_rows(T, 3 * d, x.device))
                ops.rope_qk_(qkv, llm.cos, llm.sin, S, H, D)
        q5 = qkv.unflatten(0, (B, S)).unflatten(2, (3, H, D))
        if "o_x" in lw:
            ax, a2, t2 = _ext_rows(T, d, x.device)
            attn, lse = ops.attention_fwd_lse(q5[:, :, 0], q5[:, :, 1], q5[:, :, 2], causal=True, key_valid=key_valid, out=a2.unflatten(0, (B, S)))
            s["attnd"], s["t_o"] = _adapter_down(lora, pad["o"], a2, t2, seed + 3), t2
            x_mid = ops.gemm(ax, lw["o_x"], residual=x)
        else:
            attn, lse = ops.attention_fwd_lse(q5[:, :, 0], q5[:, :, 1], q5[:, :, 2], causal=True, key_valid=key_valid)
            x_mid = ops.gemm(attn.view(T, d), lw["o"], residual=x)
        if "gu_x" in lw:
            h2x, h2, t3 = _ext_rows(T, d, x.device)
            ops.rmsnorm(x_mid, lw["ln2"], cfg.rms_norm_eps, out=h2)
        else:
            h2 = ops.rmsnorm(x_mid, lw["ln2"], cfg.rms_norm_eps)
        s.update(qkv=qkv, attn=attn, lse=lse, x_mid=x_mid, seed=seed)
        if i in llm.moe_layers:
            x_out, l_aux = (_moe_fwd_ep if llm.ep is not None else _moe_fwd)(llm, lora, i, lw, pad, h2, x_mid, s, seed)
            aux.append(l_aux)
        elif (_PRUNE_ROWS and needed is not None and i == len(llm.layers) - 1 and _SWIGLU_KEEP and (i, "ln2") not in lora.norm_names):
            # The LAST layer's MLP on the rows something reads (model_forward: supervised rows + <SEG> rows, ~1 % of the tokens): the MLP is
            # row-wise, every read row gets the result of the same arithmetic on a compact [n, d] tensor (the lora_dropout mask is a function of
            # the position inside the tensor it is drawn for, so it is another sample of the same distribution than the full-size pass would
            # have drawn; forward and backward use the same one).  x_mid's read rows are replaced in place: it becomes this layer's output.
            rows = needed[0]
            n_r = rows.numel()
            llm.pruned_rows = int(n_r)                              # the pruning HAPPENED (model_forward reports only this)
            if "gu_x" in lw:
                h2x_c, h2_c, t3_c = _ext_rows(n_r, d, x.device)
                ops.gather_rows_bf16(h2, rows, out=h2_c)
                s["h2d"], s["t_gu"] = _adapter_down(lora, pad["gu"], h2_c, t3_c, seed), t3_c
                gin, gw = h2x_c, lw["gu_x"]
            else:
                gin, gw = ops.gather_rows_bf16(h2, rows), lw["gu"]
            if "down_x" in lw:
                actx, act, t4 = _ext_rows(n_r, cfg.intermediate_size, x.device)
            else:
                act = None
            act, gu = ops.gemm_swiglu_keep(gin, gw, act_out=act)
            xm_c = ops.gather_rows_bf16(x_mid, rows)
            if "down_x" in lw:
                s["actd"], s["t_d"] = _adapter_down(lora, pad["down"], act, t4, seed + 1), t4
                out_c = ops.gemm(actx, lw["down_x"], residual=xm_c)
            else:
                out_c = ops.gemm(act, lw["down"], residual=xm_c)
            x_out = ops.scatter_rows_bf16_(x_mid, rows, out_c)
            s.update(gu=gu, rows_last=rows, xm_c=xm_c, x_mid=None)
        else:
            # gate|up: silu(gate) * up from the GEMM's epilogue, which also stores the gate|up values the backward reads
            if "down_x" in lw:
                actx, act, t4 = _ext_rows(T, cfg.intermediate_size, x.device)
            else:
                act = None
            if "gu_x" in lw:
                s["h2d"], s["t_gu"] = _adapter_down(lora, pad["gu"], h2, t3, seed), t3
                gin, gw = h2x, lw["gu_x"]
            else:
                gin, gw = h2, lw["gu"]
            if _SWIGLU_KEEP:
                act, gu = ops.gemm_swiglu_keep(gin, gw, act_out=act)
            else:                                                  # A/B: the two-kernel form
                gu = ops.gemm(gin, gw)
                act = ops.swiglu_pair_fwd(gu, out=act)
            if "down_x" in lw:
                s["actd"], s["t_d"] = _adapter_down(lora, pad["down"], act, t4, seed + 1), t4
                x_out = ops.gemm(actx, lw["down_x"], residual=x_mid)
            else:
                x_out = ops.gemm(act, lw["down"], residual=x_mid)
            s["gu"] = gu
        saved.append(s)
        x = x_out
    out = ops.rmsnorm(x, llm.norm_w, cfg.rms_norm_eps)
    aux_sum = torch.stack([a.reshape(()) for a in aux]).sum().reshape(1) if aux else torch.zeros(1, dtype=torch.float32, device=x.device)
    return out.view(B, S, d), aux_sum, {"layers": saved, "x_last": x, "B": B, "S": S, "key_valid": key_valid}


def _adapter_bwd(lora, ops_pad, dy, x, t, dx, seed, swiglu_gu=None, partials=False, defer_up=False, done=None, side_run=None):
    """Gradients of one (fused) adapter: dB_pad [out, R], dA^T [in, R] (fp32) and dx += scaling * ((dy B) A) (through the dropout).  x = the
    adapter's UNdropped input; the mask is regenerated from the seed wherever it is needed.  dx = None: nothing trainable lies in front of
    this adapter's input (the lowest layer of a decoder whose input rows are frozen) — only the two weight gradients are produced."""
    A, AT, B, BT, R, _ = ops_pad
    # [T, 64] = scaling * dy B: the down-projection kernel with B^T as its matrix (reads dy once; no dropout on this side)
    # partials: the chunk partials are handed on unsummed (ops.SkinnyPartial) — the gradient unpack into the flat buffer adds them up itself
    partials = partials and R <= 32
    if done is not None:
        dB, dt = done                                          # the kernel that produced dy took both products of it on the way (ops.swiglu_bwd_skinny)
    elif _FUSE_DY and R <= 32 and dy.stride(0) % 8 == 0 and dy.shape[1] % 8 == 0:
        dB, dt = ops.tn_skinny_down(dy, t, BT, R, lora.scaling, lora.scaling, reduce=not partials)       # both products of dy in one pass over it
    else:
        dt = ops.lora_down(dy, BT, torch.empty((dy.shape[0], 64), dtype=torch.bfloat16, device=dy.device), R, alpha=lora.scaling)
        dB = ops.tn_skinny(dy, t, R, lora.scaling, reduce=not partials)                 # [out, R] = scaling * dy^T t
    kb = lora.keep_bits.get(seed)                              # the forward's mask bytes (None: regenerate from the seed)
    # [in, R] = dropout(x)^T (scaling * dy B): nothing on the dgrad chain reads it (side_run: backward()'s weight-gradient stream, or None)
    if side_run is not None:
        dAT = side_run(lambda: ops.tn_skinny(x, dt, R, 1.0, lora.p_active, seed, reduce=not partials, keep_bits=kb), dt)
    else:
        dAT = ops.tn_skinny(x, dt, R, 1.0, lora.p_active, seed, reduce=not partials, keep_bits=kb)
    if dx is None:
        return None, dB, dAT
    if swiglu_gu is not None and R <= 32 and dx.stride(0) % 8 == 0 and _FUSE_UP_SWIGLU:
        # the adapter on down_proj: its input gradient has ONE consumer, the SwiGLU backward — both in one pass, the gate|up gradient comes back
        # (these two regenerate the mask from the seed: they are bound by their 560 MB of traffic either way — 133 us with the hash, 136 with the bytes)
        return ops.lora_up_add_swiglu_bwd(dt, AT, dx, swiglu_gu, R, lora.p_active, seed), dB, dAT
    if defer_up and R <= 16:
        # the caller's next kernel is the only reader of dx + dropout(dt A) and forms it itself (ops.rmsnorm_bwd_up): dx comes back untouched
        return (dx, (dt, AT, R, lora.p_active, seed, kb)), dB, dAT
    if R <= 32 and dx.stride(0) % 8 == 0:
        dx = ops.lora_up_add(dt, AT, dx, R, lora.p_active, seed)       # dx += dropout(dt A): the same mask and 1/(1-p) as the forward
    elif lora.p_active > 0:
        dxa = ops.dropout_bf16(ops.gemm(dt, AT), lora.p_active, seed)
        dx = ops.add3(dx, dxa)
    else:
        dx = ops.gemm(dt, AT, residual=dx)
    return dx, dB, dAT


def backward(llm, saved, d_hidden, d_aux=None, need_d_embeds=True):
    """d_hidden [B,S,d] bf16 (gradient of the stack's output), d_aux [1] fp32 (gradient of the summed l_aux) -> {parameter name: fp32 gradient}.
    need_d_embeds=False: the decoder's input rows are frozen (embed_tokens and every front-end module: the shipped stage-III configuration,
    scripts/train_stage3.sh:29-33), so in the LOWEST layer the chain stops at the last tensor a trainable parameter reads: with adapters on
    gate / up / down only that is the gate|up gradient — the layer's main gate|up dgrad GEMM, both RMSNorm backward passes, the o_proj and
    qkv dgrad GEMMs, the attention backward and the RoPE transpose would only produce the gradient of frozen embeddings (HF + peft compute
    it and drop it; ~1.7 ms of the 132 ms step)."""
    cfg, lora = llm.cfg, llm.lora
    B, S = saved["B"], saved["S"]
    H, D, d = cfg.num_attention_heads, cfg.head_dim, cfg.hidden_size
    T = B * S
    r = lora.r
    grads = {}

    def take(i, ops_pad, dB, dAT):
        """Unpack the fused pair's gradients into the per-target parameters -- straight into the parameters' .grad (the engine's flat
        buffer) when a gradient sink is attached, one launch per adapter."""
        for k, t in enumerate(ops_pad[5]):
            nb, na = f"model.layers.{i}.{_module(t)}.lora_B.default.weight", f"model.layers.{i}.{_module(t)}.lora_A.default.weight"
            pb, pa = lora.params[lora.index[nb]], lora.params[lora.index[na]]
            direct = (lora.grad_sink is not None and pb.grad is not None and pa.grad is not None and pb.grad.is_contiguous() and pa.grad.is_contiguous()
                      and pb.grad.dtype == torch.float32)
            if direct and isinstance(dB, ops.SkinnyPartial) and isinstance(dAT, ops.SkinnyPartial):
                ops.lora_grad_unpack_partials(dB, dAT, lora.rows[t], k * r, pb.grad, pa.grad)
                continue
            if isinstance(dB, ops.SkinnyPartial):
                dB = dB.finish()
            if isinstance(dAT, ops.SkinnyPartial):
                dAT = dAT.finish()
            if direct and dB.is_contiguous() and dAT.is_contiguous():
                ops.lora_grad_unpack(dB, dAT, lora.rows[t], k * r, pb.grad, pa.grad)
            else:
                grads[nb] = dB[lora.rows[t], k * r:(k + 1) * r]
                grads[na] = dAT[:, k * r:(k + 1) * r].t()

    def take_e(i, ops_pad, dB, dAT):
        """The same for the per-expert adapters of a MoE layer; dB / dAT: a list over all experts, or {global expert id: gradient} with
        the experts this rank owns (expert parallelism)."""
        for k, t in enumerate(ops_pad[5]):
            for e, mod in enumerate(lora._modules_of(i, t)):
                if isinstance(dB, dict) and e not in dB:
                    continue
                nb, na = f"model.layers.{i}.{mod}.lora_B.default.weight", f"model.layers.{i}.{mod}.lora_A.default.weight"
                pb, pa = lora.params[lora.index[nb]], lora.params[lora.index[na]]
                if (lora.grad_sink is not None and pb.grad is not None and pa.grad is not None and pb.grad.is_contiguous() and pa.grad.is_contiguous()
                        and pb.grad.dtype == torch.float32 and dB[e].is_contiguous() and dAT[e].is_contiguous()):
                    ops.lora_grad_unpack(dB[e], dAT[e], lora.rows[t], k * r, pb.grad, pa.grad)
                else:
                    grads[nb] = dB[e][lora.rows[t], k * r:(k + 1) * r]
                    grads[na] = dAT[e][:, k * r:(k + 1) * r].t()

    part_ok = lora.grad_sink is not None and _UNPACK_PARTIALS and lora.r <= 32      # the unpack into the flat gradient buffer sums the chunk partials itself
    # MP_LORA_WGRAD_STREAM=1: the dense layers' pure weight-gradient work (dA^T = drop(x)^T dt and the unpack into the flat gradient buffer: two
    # skinny products and two or three unpack launches per layer, ~2.8 ms of a 120 ms step that nothing on the dgrad chain waits for) goes to a
    # side stream; same kernels, same operands, same sums — only the queue differs.  The stream waits for the producing stream at every hand-over
    # and the producing stream waits for it before anything reads the gradients (a layer's sink, the end of backward).
    side = None
    if _WGRAD_STREAM and d_hidden.is_cuda and lora.grad_sink is not None:
        if lora.wgrad_stream is None:
            lora.wgrad_stream = torch.cuda.Stream(device=d_hidden.device)
        side = lora.wgrad_stream

    def wg(fn, *used):
        if side is None:
            return fn()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            r = fn()
        for t in used:                                             # allocated on the producing stream, read on this one
            t = t.partial if isinstance(t, ops.SkinnyPartial) else t
            if torch.is_tensor(t):
                t.record_stream(side)
        return r

    def sink(i):
        pre = f"model.layers.{i}."
        ng = {n: grads.pop(n) for n in [k for k in grads if k.startswith(pre)]}
        if side is not None and (ng or getattr(lora, "sink_reduces", True)):
            torch.cuda.current_stream().wait_stream(side)          # the layer's gradients are complete before anything adds to or reduces them
            for g in ng.values():                                  # (side-stream allocations read on this stream from here on)
                g.record_stream(torch.cuda.current_stream())
        lora.grad_sink(i, ng)

    dx = ops.rmsnorm_bwd(saved["x_last"], llm.norm_w, d_hidden.reshape(T, d).contiguous(), cfg.rms_norm_eps)
    for i in range(len(llm.layers) - 1, -1, -1):
        lw, s = llm.layers[i], saved["layers"][i]
        pad = s["pad"]
        # ---- MLP: x_out = x_mid + down(act) [+ adapter], or the MoE layer
        if s.get("moe"):
            d_h2 = (_moe_bwd_ep if s.get("ep") else _moe_bwd)(llm, lora, i, lw, s, dx, d_aux, grads, take_e)
        else:
            rows_last = s.get("rows_last")
            dy_mlp = dx if rows_last is None else ops.gather_rows_bf16(dx, rows_last)      # the pruned last layer: its MLP saw these rows only
            d_act = ops.gemm(dy_mlp, lw["down_T"])
            d_gu = None
            gu_done = None                                      # (dB, dt) of the gate|up adapter when the one-kernel form below produced them
            if ("down" in pad and "gu" in pad and _FUSE_SWSK and _FUSE_UP_SWIGLU and _FUSE_DY and pad["down"][4] <= 16 and pad["gu"][4] <= 32
                    and d_act.stride(0) % 8 == 0 and dy_mlp.stride(0) % 8 == 0 and s["gu"].is_contiguous()):
                # both adapters: the down adapter's two products of dy, its weight gradient over act, then ONE kernel from d_act to d gate|up
                # that also takes the gate|up adapter's two products of it (the 225 MB tensor is written once and not read back)
                _, ATd, _, BTd, Rd, _ = pad["down"]
                _, _, _, BTg, Rg, _ = pad["gu"]
                sd = s["seed"] + 1
                dBd, dtd = ops.tn_skinny_down(dy_mlp, s["t_d"], BTd, Rd, lora.scaling, lora.scaling, reduce=not part_ok)
                kbd = lora.keep_bits.get(sd)
                wg(lambda: take(i, pad["down"], dBd, ops.tn_skinny(s["actd"], dtd, Rd, 1.0, lora.p_active, sd, reduce=not part_ok, keep_bits=kbd)), dBd, dtd)
                d_gu, dBg, dtg = ops.swiglu_bwd_skinny(dtd, ATd, d_act, s["gu"], Rd, lora.p_active, sd, s["t_gu"], BTg, Rg, lora.scaling, lora.scaling,
                                                       reduce=not part_ok, keep_bits=kbd)
                gu_done, d_act = (dBg, dtg), None
            elif "down" in pad:
                fused = _FUSE_UP_SWIGLU and pad["down"][4] <= 32 and d_act.stride(0) % 8 == 0
                d_act, dB, dAT = _adapter_bwd(lora, pad["down"], dy_mlp, s["actd"], s["t_d"], d_act, s["seed"] + 1, swiglu_gu=s["gu"] if fused else None, partials=part_ok,
                                              side_run=wg if side is not None else None)
                wg(lambda: take(i, pad["down"], dB, dAT), dB)
                if fused:
                    d_gu, d_act = d_act, None
            if d_gu is None:
                d_gu = ops.swiglu_pair_bwd(s["gu"], d_act)
            # the lowest layer with frozen input rows: nothing trainable reads anything in front of the gate|up input unless the attention
            # projections carry adapters or a norm of this layer trains
            stop_here = (i == 0 and not need_d_embeds and "o" not in pad and "qkv" not in pad and rows_last is None
                         and (i, "ln1") not in lora.norm_names and (i, "ln2") not in lora.norm_names)
            d_h2 = None if stop_here else ops.gemm(d_gu, lw["gu_T"])
            up_late = None
            if "gu" in pad:
                # the adapter's input gradient has one reader, the post-attention norm's backward below: that kernel adds it on its way in
                defer = (_FUSE_NORM_UP and not stop_here and rows_last is None and d == 4096 and (i, "ln2") not in lora.norm_names
                         and pad["gu"][4] <= 16 and d_h2.stride(0) % 8 == 0)
                d_h2, dB, dAT = _adapter_bwd(lora, pad["gu"], d_gu, s["h2d"], s["t_gu"], d_h2, s["seed"], partials=part_ok, defer_up=defer, done=gu_done,
                                             side_run=wg if side is not None else None)
                if defer:
                    d_h2, up_late = d_h2
                wg(lambda: take(i, pad["gu"], dB, dAT), dB)
            if stop_here:
                dx = None
                if lora.grad_sink is not None:
                    sink(i)
                break
        if s.get("rows_last") is not None:
            # compact rows back into the layer's full gradient: every other row of dx is zero and stays zero (no MLP branch, no residual)
            d_mid_c = ops.rmsnorm_bwd(s["xm_c"], lw["ln2"], d_h2, cfg.rms_norm_eps, add=dy_mlp)
            d_mid = ops.scatter_rows_bf16_(dx, s["rows_last"], d_mid_c)
        elif (i, "ln2") in lora.norm_names:
            d_mid, grads[lora.norm_names[(i, "ln2")]] = ops.rmsnorm_bwd(s["x_mid"], lw["ln2"], d_h2, cfg.rms_norm_eps, add=dx, want_wgrad=True)
        elif not s.get("moe") and up_late is not None:
            dt_, AT_, R_, p_, seed_, kb_ = up_late
            d_mid = ops.rmsnorm_bwd_up(s["x_mid"], lw["ln2"], d_h2, cfg.rms_norm_eps, dt_, AT_, R_, p_, seed_, add=dx, keep_bits=kb_)
        else:
            d_mid = ops.rmsnorm_bwd(s["x_mid"], lw["ln2"], d_h2, cfg.rms_norm_eps, add=dx)
        # ---- attention: x_mid = x + o(attn(rope(qkv(rmsnorm(x))))) [+ adapters on o and on q / k / v]
        d_attn = ops.gemm(d_mid, lw["o_T"])
        if "o" in pad:
            d_attn, dB, dAT = _adapter_bwd(lora, pad["o"], d_mid, s["attnd"], s["t_o"], d_attn, s["seed"] + 3, partials=part_ok, side_run=wg if side is not None else None)
            wg(lambda: take(i, pad["o"], dB, dAT), dB)
        q5 = s["qkv"].unflatten(0, (B, S)).unflatten(2, (3, H, D))
        # the transpose of a rotation is the rotation by -theta: the attention backward stores dq / dk rotated (same bits as mp_rope_qk_bf16 on its result)
        rot = _ROPE_IN_ATTN_BWD and D == 128 and s["attn"].stride(-2) % 8 == 0
        _, _, _, dqkv = ops.attention_bwd(q5[:, :, 0], q5[:, :, 1], q5[:, :, 2], s["attn"], d_attn.view(B, S, d), s["lse"], causal=True,
                                          key_valid=saved["key_valid"], rope=(llm.cos, llm.sin_neg) if rot else None)
        dqkv = dqkv.flatten(0, 1).flatten(1)                    # [T, 3d] view of the (row-padded) buffer
        if not rot:
            ops.rope_qk_(dqkv, llm.cos, llm.sin_neg, S, H, D)
        d_h1 = ops.gemm(dqkv, lw["qkv_T"])
        if "qkv" in pad:
            d_h1, dB, dAT = _adapter_bwd(lora, pad["qkv"], dqkv, s["h1d"], s["t_qkv"], d_h1, s["seed"] + 2, partials=part_ok, side_run=wg if side is not None else None)
            wg(lambda: take(i, pad["qkv"], dB, dAT), dB)
        if (i, "ln1") in lora.norm_names:
            dx, grads[lora.norm_names[(i, "ln1")]] = ops.rmsnorm_bwd(s["x"], lw["ln1"], d_h1, cfg.rms_norm_eps, add=d_mid, want_wgrad=True)
        else:
            dx = ops.rmsnorm_bwd(s["x"], lw["ln1"], d_h1, cfg.rms_norm_eps, add=d_mid)
        if lora.grad_sink is not None:                              # hand layer i's finished gradients over (bucketed all-reduce)
            sink(i)
    if side is not None:
        torch.cuda.current_stream().wait_stream(side)              # before the optimizer (and the mask bytes' release below) — one wait per step
    grads["__d_embeds__"] = dx                                     # gradient of the decoder's input rows (for embed_tokens)
    lora.keep_bits = {}                                            # the step's mask bytes (T K / 8 per adapter and layer) die with its backward, not at the next forward
    return grads


class LlamaLoRAFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, llm, embeds, key_valid, *params):
        out, aux_sum, saved = forward_train(llm, embeds, key_valid)
        ctx.llm, ctx.saved = llm, saved
        return out, aux_sum

    @staticmethod
    def backward(ctx, d_hidden, d_aux):
        llm = ctx.llm
        grads = backward(llm, ctx.saved, d_hidden.contiguous(), None if d_aux is None else d_aux.contiguous(), need_d_embeds=bool(ctx.needs_input_grad[1]))
        d_emb = grads.pop("__d_embeds__").view(ctx.saved["B"], ctx.saved["S"], -1) if ctx.needs_input_grad[1] else None
        ctx.saved = None
        # one slot per parameter forward() took; None where the engine's grad sink already accumulated the gradient
        return (None, d_emb, None) + tuple((grads[n].contiguous() if n in grads else None) for n in decoder_param_names(llm.lora))


class CrossEntropyFn(torch.autograd.Function):
    """lm_head on the supervised rows + filtered mean CE (medplib_moe_llama.py:388-408) with its backward into the hidden states."""

    @staticmethod
    def forward(ctx, last_hidden, sup_rows, sup_labels, llm, lm_head_param=None):
        d = last_hidden.shape[-1]
        rows = ops.cast_to_bf16(ops.gather_rows_bf16_to_f32(last_hidden.reshape(-1, d), sup_rows))
        logits = ops.gemm(rows, llm.lm_head, out_dtype=torch.float32)
        ce = ops.mean_plus(ops.cross_entropy_rows(logits, sup_labels), 1.0)
        ctx.save_for_backward(logits, sup_rows, sup_labels, rows)
        ctx.llm, ctx.shape = llm, last_hidden.shape
        return ce

    @staticmethod
    def backward(ctx, g):
        logits, sup_rows, sup_labels, rows = ctx.saved_tensors
        llm = ctx.llm
        n, V = logits.shape
        vp = llm.lm_head_T.shape[1]
        dl = ops.ce_rows_bwd(logits, sup_labels, g.contiguous(), 1.0 / n, vp)
        d_rows = ops.gemm(dl, llm.lm_head_T, out_dtype=torch.float32)
        T = ctx.shape[0] * ctx.shape[1]
        d_w = None
        if ctx.needs_input_grad[4]:
            # d lm_head [V, d] = d_logits^T rows: the NT GEMM on the two small transposes (the supervised-row count padded to a K-tile)
            n64 = (n + 63) // 64 * 64
            dlT = torch.zeros((vp, n64), dtype=torch.bfloat16, device=dl.device); dlT[:, :n] = dl.t()
            rT = torch.zeros((rows.shape[1], n64), dtype=torch.bfloat16, device=dl.device); rT[:, :n] = rows.t()
            d_w = ops.gemm(dlT, rT, out_dtype=torch.float32)[:V].contiguous()
        return ops.scatter_rows_f32_bf16(d_rows, sup_rows, T).view(ctx.shape), None, None, None, d_w


class GatherRowsFn(torch.autograd.Function):
    """fp32 rows of the bf16 hidden states for the fp32 tail (text_hidden_fcs on the <SEG> rows), with the scatter backward."""

    @staticmethod
    def forward(ctx, last_hidden, rows):
        ctx.save_for_backward(rows)
        ctx.shape = last_hidden.shape
        return ops.gather_rows_bf16_to_f32(last_hidden.reshape(-1, last_hidden.shape[-1]), rows)

    @staticmethod
    def backward(ctx, g):
        (rows,) = ctx.saved_tensors
        return ops.scatter_rows_f32_bf16(g.contiguous(), rows, ctx.shape[0] * ctx.shape[1]).view(ctx.shape), None


class EmbedSpliceFn(torch.autograd.Function):
    """The multimodal splice (token-embedding gather + image-feature rows) with the embedding table's gradient: the rows that came
    from the table are grouped by token id on the host (the ids are host data) and summed per id in list order."""

    @staticmethod
    def forward(ctx, embed_param, llm, feats, src_dev, src_host, shape):
        import numpy as np
        ctx.llm, ctx.shape = llm, shape
        flat = src_host.reshape(-1)
        frows = np.flatnonzero((flat < 0) & (flat != np.iinfo(np.int64).min))         # rows that came from the feature block
        ctx.fidx = (torch.from_numpy(frows.astype(np.int64)).to(src_dev.device), torch.from_numpy((-1 - flat[frows]).astype(np.int64)).to(src_dev.device),
                    feats.shape[0])
        tok_rows = np.flatnonzero(flat >= 0)
        order = tok_rows[np.argsort(flat[tok_rows], kind="stable")]
        ids_sorted = flat[order]
        uniq, first = np.unique(ids_sorted, return_index=True)
        seg = np.concatenate([first, [ids_sorted.size]]).astype(np.int64)
        dev = src_dev.device
        ctx.idx = tuple(torch.from_numpy(a.astype(np.int64)).to(dev) for a in (order, seg, uniq))
        return ops.splice_rows(llm.embed_tokens, feats, src_dev, shape[2]).view(shape)

    @staticmethod
    def backward(ctx, g):
        order, seg, uniq = ctx.idx
        V, d = ctx.llm.embed_tokens.shape
        g2 = g.reshape(-1, d).contiguous()
        d_emb = ops.embed_grad(g2, order, seg, uniq, V) if ctx.needs_input_grad[0] else None
        d_feats = None
        if ctx.needs_input_grad[2]:                                 # every feature row is spliced at most once: a row copy
            rows, fi, nf = ctx.fidx
            d_feats = torch.zeros((nf, d), dtype=torch.bfloat16, device=g2.device)
            d_feats[fi] = g2[rows]
        return d_emb, None, d_feats, None, None, None


class ProjectorFn(torch.autograd.Function):
    """mm_projector (Linear - GELU - Linear, multimodal_projector/builder.py:39-46) with trainable weights: the tower features are
    frozen inputs; the weight gradients are NT GEMMs on transposed copies of the (small) token-major operands."""

    @staticmethod
    def forward(ctx, raw, tower, w0, b0, w2, b2):
        pr = tower.proj
        h_pre = ops.gemm(raw, pr["w0"], bias=pr["b0"])
        h = ops.gelu_fwd_bf16(h_pre)
        ctx.save_for_backward(raw, h_pre, h)
        ctx.tower = tower
        return ops.gemm(h, pr["w2"], bias=pr["b2"])

    @staticmethod
    def backward(ctx, g):
        raw, h_pre, h = ctx.saved_tensors
        pr = ctx.tower.proj
        g = g.contiguous()
        n = g.shape[0]
        n64 = (n + 63) // 64 * 64

        def tpad(x):                                                # [n, c] -> [c, n64] (zero columns beyond n)
            out = torch.zeros((x.shape[1], n64), dtype=torch.bfloat16, device=x.device)
            out[:, :n] = x.t()
            return out
        gT = tpad(g)
        d_w2 = ops.gemm(gT, tpad(h), out_dtype=torch.float32)                       # [out, in] = g^T h
        d_b2 = ops.colsum_f32(ops.cast_to_f32(g))
        d_hpre = ops.gelu_bwd_bf16(h_pre, ops.gemm(g, pr["w2_T"]))
        d_w0 = ops.gemm(tpad(d_hpre), tpad(raw), out_dtype=torch.float32)
        d_b0 = ops.colsum_f32(ops.cast_to_f32(d_hpre))
        return None, None, d_w0, d_b0, d_w2, d_b2



def _tpad(x):
    """[n, c] bf16 -> [c, n64] with zero columns beyond n: the operand shape of the NT GEMM for x^T-products."""
    n = x.shape[0]
    out = torch.zeros((x.shape[1], (n + 63) // 64 * 64), dtype=torch.bfloat16, device=x.device)
    out[:, :n] = x.t()
    return out


class TokenCompressorFn(torch.autograd.Function):
    """TokenCompressor (AdaptiveAvgPool1d over tokens -> LayerNorm -> Linear, medplib_arch.py:67-77) with trainable LayerNorm and
    Linear; its input (the projector's output) is frozen."""

    @staticmethod
    def forward(ctx, feats, comp, n_images, tokens_in, nw, nb, pw, pb):
        pooled = ops.adaptive_avgpool_tokens(feats.view(n_images, tokens_in, comp.hidden), comp.num_tokens).view(-1, comp.hidden)
        h = ops.layernorm(pooled, comp.norm[0], comp.norm[1], 1e-5)
        ctx.save_for_backward(pooled, h)
        ctx.comp = comp
        return ops.gemm(h, comp.proj_w, bias=comp.proj_b)

    @staticmethod
    def backward(ctx, g):
        pooled, h = ctx.saved_tensors
        comp = ctx.comp
        g = g.contiguous()
        d_pw = ops.gemm(_tpad(g), _tpad(h), out_dtype=torch.float32)
        d_pb = ops.colsum_f32(ops.cast_to_f32(g))
        d_h = ops.cast_to_f32(ops.gemm(g, comp.proj_w_T))
        xf = ops.cast_to_f32(pooled)
        _, mean, rstd = ops.layernorm_fwd_f32(xf, comp.norm[0], comp.norm[1], 1e-5)
        dw, db = torch.zeros_like(comp.norm[0]), torch.zeros_like(comp.norm[0])
        ops.layernorm_bwd_f32(d_h, xf, comp.norm[0], mean, rstd, dw, db)
        return None, None, None, None, dw, db, d_pw, d_pb


class RegionAdapterFn(torch.autograd.Function):
    """region_fea_adapter (Linear on the raw tower features) + extract_region_feature (medplib_arch.py:207, 580-613) with a trainable
    adapter: the point-sampling mean's backward spreads each mask's gradient over the pixels its points touch (gather form), the
    Linear's weight gradient is an NT GEMM on transposed copies."""

    @staticmethod
    def forward(ctx, raw_sel, tower, xy, offsets, map_index, hw, w, b):
        ra = tower.region_adapter
        n_maps = raw_sel.shape[0] // (hw * hw)
        fmap = ops.gemm(raw_sel, ra["w"], bias=ra["b"]).view(n_maps, hw * hw, -1)
        ctx.save_for_backward(raw_sel, xy, offsets, map_index)
        ctx.hw, ctx.n_maps = hw, n_maps
        return ops.region_point_mean(fmap.contiguous(), xy, offsets, map_index, hw, hw)

    @staticmethod
    def backward(ctx, g):
        raw_sel, xy, offsets, map_index = ctx.saved_tensors
        dfmap = ops.region_point_mean_bwd(xy, offsets, map_index, g.contiguous(), ctx.n_maps, ctx.hw, ctx.hw)
        dfm2 = dfmap.view(-1, dfmap.shape[-1])
        d_w = ops.gemm(_tpad(dfm2), _tpad(raw_sel), out_dtype=torch.float32)
        d_b = ops.colsum_f32(ops.cast_to_f32(dfm2))
        return None, None, None, None, None, None, d_w, d_b


class MaskEncoderFn(torch.autograd.Function):
    """MaskTokenEncoder (4 x [Conv3x3 s2 p1 + GELU] -> AdaptiveAvgPool1d over tokens -> Linear -> LayerNorm, medplib_arch.py:80-108)
    with every parameter trainable.  The training forward keeps the pre-activations (unfused GELU) and the im2col matrices; the
    backward is dgrad GEMM + gather-form col2im per convolution, NT GEMMs on transposed copies for the weight gradients."""

    @staticmethod
    def forward(ctx, masks, enc, *params):
        from .icl import _conv_taps
        if masks.dim() == 4:
            masks = masks[:, 0]
        if masks.dtype not in (torch.float32, torch.bfloat16):
            masks = masks.float()
        masks = masks.contiguous()
        pre = [ops.conv3x3s2_c1_pre(masks, enc.w0, enc.b0)]
        x = ops.gelu_fwd_bf16(pre[0])
        n = x.shape[0]
        cols, shapes = [], []
        for w, b in enc.convs:
            H, W, C = x.shape[1:]
            OH, OW = (H + 2 - 3) // 2 + 1, (W + 2 - 3) // 2 + 1
            shapes.append((H, W, C, OH, OW))
            cols.append(ops.im2col_nhwc(x, OH, OW, 2, _conv_taps(3, 1)))
            pre.append(ops.gemm(cols[-1], w, bias=b).view(n, OH, OW, w.shape[0]))
            x = ops.gelu_fwd_bf16(pre[-1])
        L = x.shape[1] * x.shape[2]
        toks = ops.adaptive_avgpool_tokens(x.view(n, L, x.shape[3]), enc.num_tokens).view(-1, 256)
        h = ops.gemm(toks, enc.proj_w, bias=enc.proj_b)
        ctx.enc, ctx.masks, ctx.pre, ctx.cols, ctx.shapes, ctx.toks, ctx.h, ctx.L, ctx.n = enc, masks, pre, cols, shapes, toks, h, L, n
        return ops.layernorm(h, enc.norm[0], enc.norm[1], 1e-5)

    @staticmethod
    def backward(ctx, g):
        enc, n = ctx.enc, ctx.n
        hf = ops.cast_to_f32(ctx.h)
        _, mean, rstd = ops.layernorm_fwd_f32(hf, enc.norm[0], enc.norm[1], 1e-5)
        d_nw, d_nb = torch.zeros_like(enc.norm[0]), torch.zeros_like(enc.norm[0])
        d_h = ops.cast_to_bf16(ops.layernorm_bwd_f32(ops.cast_to_f32(g.contiguous()), hf, enc.norm[0], mean, rstd, d_nw, d_nb))
        d_pw = ops.gemm(_tpad(d_h), _tpad(ctx.toks), out_dtype=torch.float32)
        d_pb = ops.colsum_f32(ops.cast_to_f32(d_h))
        d_toks = ops.gemm(d_h, enc.proj_w.t().contiguous())
        d_x = ops.adaptive_avgpool_tokens_bwd(d_toks.view(n, enc.num_tokens, 256), ctx.L)
        conv_grads = []
        for j in (2, 1, 0):
            w, _ = enc.convs[j]
            H, W, C, OH, OW = ctx.shapes[j]
            d_pre = ops.gelu_bwd_bf16(ctx.pre[j + 1].view(-1, w.shape[0]), d_x.reshape(-1, w.shape[0]).contiguous())
            conv_grads.append((ops.gemm(_tpad(d_pre), _tpad(ctx.cols[j]), out_dtype=torch.float32), ops.colsum_f32(ops.cast_to_f32(d_pre))))
            d_cols = ops.gemm(d_pre, w.t().contiguous())
            d_x = ops.col2im_k3s2p1(d_cols, n, H, W, C)
        d_pre1 = ops.gelu_bwd_bf16(ctx.pre[0].view(-1, 64), d_x.reshape(-1, 64).contiguous()).view(ctx.pre[0].shape)
        d_w0, d_b0 = ops.conv3x3s2_c1_wgrad(ctx.masks, d_pre1)
        out = [d_w0, d_b0]
        for dw, db in reversed(conv_grads):
            out += [dw, db]
        return (None, None) + tuple(out) + (d_pw, d_pb, d_nw, d_nb)
