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
fp32 tail); everything inside them is this library's kernels.  Targets: any of q/k/v/o/gate/up/down_proj (the shipped scripts' sets);
adapters inside MoE layers are not built yet."""
import math

import torch

from .. import ops

MLP_TARGETS = ("gate_proj", "up_proj", "down_proj")
ALL_TARGETS = ("q_proj", "k_proj", "v_proj", "o_proj") + MLP_TARGETS
# adapter groups: targets that share an input and whose outputs are row blocks of one fused projection get ONE pair of thin GEMMs
GROUPS = {"qkv": ("q_proj", "k_proj", "v_proj"), "o": ("o_proj",), "gu": ("gate_proj", "up_proj"), "down": ("down_proj",)}


def _module(t):
    return ("self_attn." if t in ("q_proj", "k_proj", "v_proj", "o_proj") else "mlp.") + t


class LoRAState(torch.nn.Module):
    """The adapters of every decoder layer as fp32 nn.Parameters (the engine's flat AdamW buffer adopts them) plus the bf16 padded
    GEMM operands rebuilt from them before each forward."""

    def __init__(self, cfg, llm, r=8, alpha=16, dropout=0.0, targets=MLP_TARGETS, seed=0):
        super().__init__()
        assert r % 8 == 0 and 0 < r <= 16, "lora_r must be 8 or 16 (the shipped scripts' values)"
        assert all(t in ALL_TARGETS for t in targets), f"adapters are built for {ALL_TARGETS}"
        assert not llm.moe_layers, "LoRA training is built for the dense decoder (MoE layers + adapters: not yet)"
        self.r, self.alpha, self.p = r, float(alpha), float(dropout)
        self.targets = tuple(t for t in ALL_TARGETS if t in targets)
        self.scaling = self.alpha / r
        d, ff, dev = cfg.hidden_size, cfg.intermediate_size, llm.device
        g = torch.Generator().manual_seed(seed)
        self.names, plist = [], []
        for i in range(cfg.num_hidden_layers):
            for t in self.targets:
                fin, fout = (ff, d) if t == "down_proj" else ((d, ff) if t in ("gate_proj", "up_proj") else (d, d))
                bound = 1.0 / math.sqrt(fin)                      # kaiming_uniform_(a=sqrt(5)) on [r, in]
                a = (torch.rand(r, fin, generator=g) * 2 - 1) * bound
                self.names += [f"model.layers.{i}.{_module(t)}.lora_A.default.weight", f"model.layers.{i}.{_module(t)}.lora_B.default.weight"]
                plist += [torch.nn.Parameter(a.to(dev)), torch.nn.Parameter(torch.zeros(fout, r, device=dev))]
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
        self._bufs = {}

    def get(self, i, t, which):
        return self.params[self.index[f"model.layers.{i}.{_module(t)}.lora_{which}.default.weight"]]

    def peft_state_dict(self):
        """The adapters under peft's key names (`base_model.model.<module>.lora_{A,B}.default.weight`), bf16 like a bf16 peft model saves
        them; `lora.merge_lora_state_dict` folds such a dict into the base weights (merge_lora_weights_and_save_hf_model_moe.py:270-345)."""
        return {"base_model.model." + n: p.detach().to(torch.bfloat16) for n, p in zip(self.names, self.params)}

    def load_peft_state_dict(self, sd):
        for n, p in zip(self.names, self.params):
            k = n if n in sd else "base_model.model." + n
            p.data.copy_(sd[k].to(p.dtype))

    def padded(self, i):
        """bf16 GEMM operands of layer i per adapter group: (A [64, in], A^T [in, 64], B [out, 64], B^T [64, out], R, targets).  The
        buffers persist (zeroed once: the padding never changes); each step one pack kernel per adapter rewrites its slices.
        R = rank of the fused pair (targets x r), rounded up to what the wgrad kernel takes."""
        r, dev, bf = self.r, self.rows["q_proj"].device, torch.bfloat16
        if i not in self._bufs:
            self._bufs[i] = {}
            for grp, members in GROUPS.items():
                tg = [t for t in members if t in self.targets]
                if tg:
                    fin, W = self.get(i, tg[0], "A").shape[1], self.width[grp]
                    self._bufs[i][grp] = (torch.zeros(64, fin, dtype=bf, device=dev), torch.zeros(fin, 64, dtype=bf, device=dev),
                                          torch.zeros(W, 64, dtype=bf, device=dev), torch.zeros(64, W, dtype=bf, device=dev), tg)
        out = {}
        for grp, (A, AT, B, BT, tg) in self._bufs[i].items():
            for k, t in enumerate(tg):
                ops.lora_pack(self.get(i, t, "A").detach(), self.get(i, t, "B").detach(), self.rows[t], A, AT, B, BT, k * r)
            R = len(tg) * r
            out[grp] = (A, AT, B, BT, 8 if R <= 8 else 16 if R <= 16 else 32 if R <= 32 else 64, tg)
        return out


def enable_lora(llm, cfg, r=8, alpha=16, dropout=0.0, targets=MLP_TARGETS, seed=0):
    """Attach adapters to a LlamaStack and make the transposed weight copies the dgrad GEMMs read."""
    llm.lora = LoRAState(cfg, llm, r, alpha, dropout, targets, seed)
    for lw in llm.layers:
        for k in ("qkv", "o", "gu", "down"):
            lw[k + "_T"] = lw[k].t().contiguous()
    V, d = llm.lm_head.shape
    vp = (V + 63) // 64 * 64
    llm.lm_head_T = torch.zeros(d, vp, dtype=torch.bfloat16, device=llm.device)
    llm.lm_head_T[:, :V] = llm.lm_head.t()
    llm.sin_neg = (-llm.sin).contiguous()
    return llm.lora


def _adapter_fwd(lora, ops_pad, x, y, seed):
    """y + scaling * (dropout(x) A^T) B^T  -> (y', x_dropped, t)."""
    A, _, B, _, _, _ = ops_pad
    xd = ops.dropout_bf16(x, lora.p, seed) if lora.p > 0 else x
    t = ops.gemm(xd, A)                                           # [T, 64] (columns >= R are zero)
    return ops.gemm(t, B, residual=y, alpha=lora.scaling), xd, t


def forward_train(llm, embeds, key_valid):
    """The decoder forward in training-with-adapters mode -> (last_hidden [B,S,d], saved)."""
    cfg, lora = llm.cfg, llm.lora
    B, S, d = embeds.shape
    H, D = cfg.num_attention_heads, cfg.head_dim
    T = B * S
    x = embeds.reshape(T, d)
    lora.step += 1
    saved = []
    for i, lw in enumerate(llm.layers):
        pad = lora.padded(i)
        s = {"x": x, "pad": pad}
        seed = (lora.step * 4096 + i) * 4
        h1 = ops.rmsnorm(x, lw["ln1"], cfg.rms_norm_eps)
        qkv = ops.gemm(h1, lw["qkv"])
        if "qkv" in pad:
            qkv, s["h1d"], s["t_qkv"] = _adapter_fwd(lora, pad["qkv"], h1, qkv, seed + 2)
        ops.rope_qk_(qkv, llm.cos, llm.sin, S, H, D)
        q5 = qkv.view(B, S, 3, H, D)
        attn, lse = ops.attention_fwd_lse(q5[:, :, 0], q5[:, :, 1], q5[:, :, 2], causal=True, key_valid=key_valid)
        x_mid = ops.gemm(attn.view(T, d), lw["o"], residual=x)
        if "o" in pad:
            x_mid, s["attnd"], s["t_o"] = _adapter_fwd(lora, pad["o"], attn.view(T, d), x_mid, seed + 3)
        h2 = ops.rmsnorm(x_mid, lw["ln2"], cfg.rms_norm_eps)
        gu = ops.gemm(h2, lw["gu"])
        s.update(qkv=qkv, attn=attn, lse=lse, x_mid=x_mid)
        if "gu" in pad:
            gu, s["h2d"], s["t_gu"] = _adapter_fwd(lora, pad["gu"], h2, gu, seed)
        act = ops.swiglu_pair_fwd(gu)
        x_out = ops.gemm(act, lw["down"], residual=x_mid)
        if "down" in pad:
            x_out, s["actd"], s["t_d"] = _adapter_fwd(lora, pad["down"], act, x_out, seed + 1)
        s["gu"], s["seed"] = gu, seed
        saved.append(s)
        x = x_out
    out = ops.rmsnorm(x, llm.norm_w, cfg.rms_norm_eps)
    return out.view(B, S, d), {"layers": saved, "x_last": x, "B": B, "S": S, "key_valid": key_valid}


def _adapter_bwd(lora, ops_pad, dy, xd, t, dx, seed):
    """Gradients of one (fused) adapter: dB_pad [out, R], dA^T [in, R] (fp32) and dx += scaling * ((dy B) A) (through the dropout)."""
    A, AT, B, BT, R, _ = ops_pad
    dt = ops.gemm(dy, BT, alpha=lora.scaling)                      # [T, 64] = scaling * dy B
    dB = ops.tn_skinny(dy, t, R, lora.scaling)                     # [out, R] = scaling * dy^T t
    dAT = ops.tn_skinny(xd, dt, R, 1.0)                            # [in, R]  = x_d^T (scaling * dy B)
    if lora.p > 0:
        dxa = ops.dropout_bf16(ops.gemm(dt, AT), lora.p, seed)    # the same mask and 1/(1-p) as the forward
        dx = ops.add3(dx, dxa)
    else:
        dx = ops.gemm(dt, AT, residual=dx)
    return dx, dB, dAT


def backward(llm, saved, d_hidden):
    """d_hidden [B,S,d] bf16 (gradient of the stack's output) -> {parameter name: fp32 gradient}."""
    cfg, lora = llm.cfg, llm.lora
    B, S = saved["B"], saved["S"]
    H, D, d = cfg.num_attention_heads, cfg.head_dim, cfg.hidden_size
    T = B * S
    r = lora.r
    grads = {}

    def take(i, ops_pad, dB, dAT):
        """Unpack the fused pair's gradients into the per-target parameters."""
        for k, t in enumerate(ops_pad[5]):
            grads[f"model.layers.{i}.{_module(t)}.lora_B.default.weight"] = dB[lora.rows[t], k * r:(k + 1) * r]
            grads[f"model.layers.{i}.{_module(t)}.lora_A.default.weight"] = dAT[:, k * r:(k + 1) * r].t()

    dx = ops.rmsnorm_bwd(saved["x_last"], llm.norm_w, d_hidden.reshape(T, d).contiguous(), cfg.rms_norm_eps)
    for i in range(len(llm.layers) - 1, -1, -1):
        lw, s = llm.layers[i], saved["layers"][i]
        pad = s["pad"]
        # ---- MLP: x_out = x_mid + down(act) [+ adapter]
        d_act = ops.gemm(dx, lw["down_T"])
        if "down" in pad:
            d_act, dB, dAT = _adapter_bwd(lora, pad["down"], dx, s["actd"], s["t_d"], d_act, s["seed"] + 1)
            take(i, pad["down"], dB, dAT)
        d_gu = ops.swiglu_pair_bwd(s["gu"], d_act)
        d_h2 = ops.gemm(d_gu, lw["gu_T"])
        if "gu" in pad:
            d_h2, dB, dAT = _adapter_bwd(lora, pad["gu"], d_gu, s["h2d"], s["t_gu"], d_h2, s["seed"])
            take(i, pad["gu"], dB, dAT)
        d_mid = ops.rmsnorm_bwd(s["x_mid"], lw["ln2"], d_h2, cfg.rms_norm_eps, add=dx)
        # ---- attention: x_mid = x + o(attn(rope(qkv(rmsnorm(x))))) [+ adapters on o and on q / k / v]
        d_attn = ops.gemm(d_mid, lw["o_T"])
        if "o" in pad:
            d_attn, dB, dAT = _adapter_bwd(lora, pad["o"], d_mid, s["attnd"], s["t_o"], d_attn, s["seed"] + 3)
            take(i, pad["o"], dB, dAT)
        q5 = s["qkv"].view(B, S, 3, H, D)
        _, _, _, dqkv = ops.attention_bwd(q5[:, :, 0], q5[:, :, 1], q5[:, :, 2], s["attn"], d_attn.view(B, S, d), s["lse"], causal=True,
                                          key_valid=saved["key_valid"])
        dqkv = dqkv.view(T, 3 * d)
        ops.rope_qk_(dqkv, llm.cos, llm.sin_neg, S, H, D)           # the transpose of a rotation is the rotation by -theta
        d_h1 = ops.gemm(dqkv, lw["qkv_T"])
        if "qkv" in pad:
            d_h1, dB, dAT = _adapter_bwd(lora, pad["qkv"], dqkv, s["h1d"], s["t_qkv"], d_h1, s["seed"] + 2)
            take(i, pad["qkv"], dB, dAT)
        dx = ops.rmsnorm_bwd(s["x"], lw["ln1"], d_h1, cfg.rms_norm_eps, add=d_mid)
    return grads


class LlamaLoRAFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, llm, embeds, key_valid, *params):
        out, saved = forward_train(llm, embeds, key_valid)
        ctx.llm, ctx.saved = llm, saved
        return out

    @staticmethod
    def backward(ctx, d_hidden):
        llm = ctx.llm
        grads = backward(llm, ctx.saved, d_hidden.contiguous())
        ctx.saved = None
        return (None, None, None) + tuple(grads[n].contiguous() for n in llm.lora.names)


class CrossEntropyFn(torch.autograd.Function):
    """lm_head on the supervised rows + filtered mean CE (medplib_moe_llama.py:388-408) with its backward into the hidden states."""

    @staticmethod
    def forward(ctx, last_hidden, sup_rows, sup_labels, llm):
        d = last_hidden.shape[-1]
        rows = ops.cast_to_bf16(ops.gather_rows_bf16_to_f32(last_hidden.reshape(-1, d), sup_rows))
        logits = ops.gemm(rows, llm.lm_head, out_dtype=torch.float32)
        ce = ops.mean_plus(ops.cross_entropy_rows(logits, sup_labels), 1.0)
        ctx.save_for_backward(logits, sup_rows, sup_labels)
        ctx.llm, ctx.shape = llm, last_hidden.shape
        return ce

    @staticmethod
    def backward(ctx, g):
        logits, sup_rows, sup_labels = ctx.saved_tensors
        llm = ctx.llm
        n = logits.shape[0]
        dl = ops.ce_rows_bwd(logits, sup_labels, g.contiguous(), 1.0 / n, llm.lm_head_T.shape[1])
        d_rows = ops.gemm(dl, llm.lm_head_T, out_dtype=torch.float32)
        T = ctx.shape[0] * ctx.shape[1]
        return ops.scatter_rows_f32_bf16(d_rows, sup_rows, T).view(ctx.shape), None, None, None


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
