"""Llama-7B decoder stack (dense or DeepSpeed-style MoE MLPs) over the HIP kernels.

Mirrors `MoELlamaModel_forward` / `MoELlamaDecoderLayer_forward` (model/medplib/model/language_model/
medplib_moe_llama.py:110-305) and HF-4.31 LlamaAttention/LlamaMLP/LlamaRMSNorm:
    x = x + o_proj(attn(rope(q), rope(k), v));  x = x + mlp(rmsnorm(x))   with mlp = SwiGLU or MoE(top-1 experts).
Weights are held in kernel layout (fused qkv [3d,d], gate/up fused [2ff,d] with rows interleaved in blocks of 32 for the
SwiGLU-in-epilogue GEMM, experts stacked [E,...]) and are
imported from / exported to the HF checkpoint key layout by load_hf / export_hf."""
import math

import torch

from .. import ops


def _rope_tables(seq, head_dim, theta, device):
    inv = 1.0 / (theta ** (torch.arange(0, head_dim, 2, dtype=torch.float32) / head_dim))
    freqs = torch.outer(torch.arange(seq, dtype=torch.float32), inv)
    return freqs.cos().contiguous().to(device), freqs.sin().contiguous().to(device)


class LlamaStack:
    def __init__(self, cfg, device, init_std=0.02, seed=0):
        self.cfg, self.device = cfg, device
        d, ff, L, V = cfg.hidden_size, cfg.intermediate_size, cfg.num_hidden_layers, cfg.vocab_size
        E = cfg.num_experts
        self.moe_layers = cfg.moe_layer_set()
        g = torch.Generator(device=device).manual_seed(seed)

        def rn(*shape, std=init_std, dtype=torch.bfloat16):
            return (torch.randn(*shape, generator=g, device=device, dtype=torch.float32) * std).to(dtype)
        self.embed_tokens = rn(V, d)
        self.lm_head = rn(V, d)
        self.norm_w = torch.ones(d, dtype=torch.float32, device=device)
        self.layers = []
        for i in range(L):
            lw = {"qkv": rn(3 * d, d), "o": rn(d, d), "ln1": torch.ones(d, dtype=torch.float32, device=device),
                  "ln2": torch.ones(d, dtype=torch.float32, device=device)}
            if i in self.moe_layers:
                lw["wg"] = rn(E, d, dtype=torch.float32)
                lw["gu"] = rn(E, 2 * ff, d)
                lw["down"] = rn(E, d, ff)
                if cfg.use_residual:              # MoE.mlp (a dense copy of the expert) + MoE.coefficient = Linear(hidden, 2)
                    lw["res_gu"] = rn(2 * ff, d)
                    lw["res_down"] = rn(d, ff)
                    lw["coef_w"] = torch.zeros(8, d, dtype=torch.bfloat16, device=device)     # rows 2..7 are padding
                    lw["coef_w"][:2] = rn(2, d)
                    lw["coef_b"] = torch.zeros(8, dtype=torch.float32, device=device)
            else:
                lw["gu"] = rn(2 * ff, d)
                lw["down"] = rn(d, ff)
            self.layers.append(lw)
        self.cos, self.sin = _rope_tables(cfg.max_position_embeddings, cfg.head_dim, cfg.rope_theta, device)
        self.training = True
        self.rts_uniform_provider = None   # callable(layer_idx, T, E) -> fp32 [T,E] gate draws (RTS uniforms / top-2 Gumbel) or None
        self.gate_pass = 0                 # forward passes so far: part of the key of the stateless gate-draw generator
        self.ep = None                     # ExpertParallel (expert_parallel.py) once enable_expert_parallel() sharded the experts
        self.fuse_moe_gather_scatter = True   # top-1, one rank: dispatch / combine folded into the expert GEMMs
        self.fuse_decode_routing = True       # decode rows: post-attention norm + gate + routing in one launch
        # RoPE in the qkv GEMM's epilogue (mp_gemm_qkv_rope_bf16) wants the q / k head rows interleaved in blocks of 32; the plain
        # fused qkv weight stays (decode GEMVs, the LoRA path's dgrad transposes, export), the interleaved copy (+3.2 GB of 288 at
        # 7B) serves every multi-row forward.  Rebuilt whenever the qkv weights change (refresh_fused_qkv).
        self.fuse_rope = cfg.head_dim == 128 and cfg.hidden_size % 256 == 0
        self.refresh_fused_qkv()

    def refresh_fused_qkv(self):
        """(Re)build the RoPE-interleaved copies of the fused qkv weights; call after anything that writes lw["qkv"]."""
        for lw in self.layers:
            if self.fuse_rope:
                lw["qkv_rope"] = ops.rope_interleave_qkv(lw["qkv"], self.cfg.num_attention_heads, self.cfg.head_dim)
            else:
                lw.pop("qkv_rope", None)
        self.refresh_folded_norms()

    def refresh_folded_norms(self):
        """config.fold_input_norm: the frozen weights with the norm weights multiplied into their columns — W' = bf16(W * ln_w[None, :]) — for the two
        consumer GEMMs of a top-1 MoE layer's RMSNorms (qkv + RoPE, experts' gate|up).  Call after anything that writes ln1 / ln2 / qkv / gu.
        (+ 0.46 GB per layer at 7B, E = 2: the unfolded weights stay for decode, export and the adapter path.)"""
        on = bool(getattr(self.cfg, "fold_input_norm", False)) and self.fuse_rope and self.cfg.top_k_experts == 1 and not self.cfg.use_residual
        for i, lw in enumerate(self.layers):
            lw.pop("qkv_rope_f", None); lw.pop("gu_f", None)
            if on and i in self.moe_layers:
                lw["qkv_rope_f"] = (lw["qkv_rope"].float() * lw["ln1"][None, :]).to(torch.bfloat16)
                lw["gu_f"] = (lw["gu"].float() * lw["ln2"][None, None, :]).to(torch.bfloat16)

    def enable_expert_parallel(self, ep):
        """Shard the experts over an expert-parallel group (DeepSpeed `ep_size`, medplib_moe_llama.py:604-614): every rank keeps the
        weights of its E/ep experts only; the gate `wg` stays replicated.  Call after the weights are loaded."""
        ids = ep.local_expert_ids()
        for i in self.moe_layers:
            lw = self.layers[i]
            for k in ("gu", "down", "gu_T", "down_T"):           # (+ the dgrad transposes when enable_lora() ran first)
                if k in lw:
                    lw[k] = lw[k][ids[0]:ids[-1] + 1].contiguous()
            for k in ("gu", "down"):                             # the expert-parallel training path keeps the unfused adapter kernels
                lw.pop(k + "_x", None)
                if getattr(self, "lora", None) is not None:
                    self.lora.ext.pop((i, k), None)
        self.ep = ep

    # ------------------------------------------------------------------ HF checkpoint layout
    def load_hf(self, sd, prefix=""):
        cfg = self.cfg
        d, ff = cfg.hidden_size, cfg.intermediate_size

        def put(dst, src):
            dst.copy_(src.to(device=dst.device, dtype=dst.dtype))
        put(self.embed_tokens, sd[prefix + "model.embed_tokens.weight"])
        put(self.lm_head, sd[prefix + "lm_head.weight"])
        put(self.norm_w, sd[prefix + "model.norm.weight"])
        for i, lw in enumerate(self.layers):
            p = f"{prefix}model.layers.{i}."
            for j, n in enumerate(("q", "k", "v")):
                put(lw["qkv"][j * d:(j + 1) * d], sd[p + f"self_attn.{n}_proj.weight"])
            put(lw["o"], sd[p + "self_attn.o_proj.weight"])
            put(lw["ln1"], sd[p + "input_layernorm.weight"])
            put(lw["ln2"], sd[p + "post_attention_layernorm.weight"])
            if i in self.moe_layers:
                put(lw["wg"], sd[p + "mlp.deepspeed_moe.gate.wg.weight"])
                for e in range(cfg.num_experts):
                    ep = p + f"mlp.deepspeed_moe.experts.deepspeed_experts.{e}."
                    put(lw["gu"][e], ops.swiglu_interleave(sd[ep + "gate_proj.weight"], sd[ep + "up_proj.weight"]))
                    put(lw["down"][e], sd[ep + "down_proj.weight"])
                if self.cfg.use_residual:
                    put(lw["res_gu"], ops.swiglu_interleave(sd[p + "mlp.mlp.gate_proj.weight"], sd[p + "mlp.mlp.up_proj.weight"]))
                    put(lw["res_down"], sd[p + "mlp.mlp.down_proj.weight"])
                    put(lw["coef_w"][:2], sd[p + "mlp.coefficient.weight"])
                    put(lw["coef_b"][:2], sd[p + "mlp.coefficient.bias"])
            else:
                put(lw["gu"], ops.swiglu_interleave(sd[p + "mlp.gate_proj.weight"], sd[p + "mlp.up_proj.weight"]))
                put(lw["down"], sd[p + "mlp.down_proj.weight"])
        self.refresh_fused_qkv()

    def export_hf(self, prefix=""):
        cfg = self.cfg
        d, ff = cfg.hidden_size, cfg.intermediate_size
        bf = torch.bfloat16
        sd = {prefix + "model.embed_tokens.weight": self.embed_tokens, prefix + "lm_head.weight": self.lm_head,
              prefix + "model.norm.weight": self.norm_w.to(bf)}
        for i, lw in enumerate(self.layers):
            p = f"{prefix}model.layers.{i}."
            for j, n in enumerate(("q", "k", "v")):
                sd[p + f"self_attn.{n}_proj.weight"] = lw["qkv"][j * d:(j + 1) * d]
            sd[p + "self_attn.o_proj.weight"] = lw["o"]
            sd[p + "input_layernorm.weight"] = lw["ln1"].to(bf)
            sd[p + "post_attention_layernorm.weight"] = lw["ln2"].to(bf)
            if i in self.moe_layers:
                sd[p + "mlp.deepspeed_moe.gate.wg.weight"] = lw["wg"]
                for e in range(cfg.num_experts):
                    ep = p + f"mlp.deepspeed_moe.experts.deepspeed_experts.{e}."
                    sd[ep + "gate_proj.weight"], sd[ep + "up_proj.weight"] = ops.swiglu_deinterleave(lw["gu"][e])
                    sd[ep + "down_proj.weight"] = lw["down"][e]
                if self.cfg.use_residual:
                    sd[p + "mlp.mlp.gate_proj.weight"], sd[p + "mlp.mlp.up_proj.weight"] = ops.swiglu_deinterleave(lw["res_gu"])
                    sd[p + "mlp.mlp.down_proj.weight"] = lw["res_down"]
                    sd[p + "mlp.coefficient.weight"] = lw["coef_w"][:2]
                    sd[p + "mlp.coefficient.bias"] = lw["coef_b"][:2].to(torch.bfloat16)
            else:
                sd[p + "mlp.gate_proj.weight"], sd[p + "mlp.up_proj.weight"] = ops.swiglu_deinterleave(lw["gu"])
                sd[p + "mlp.down_proj.weight"] = lw["down"]
        return sd

    # ------------------------------------------------------------------ forward
    def capacity(self, T):
        """DeepSpeed _capacity: ceil(T / E * cf), at least min_capacity; top2gating passes 2 * cf."""
        cfg = self.cfg
        cf = cfg.capacity_factor if self.training else cfg.eval_capacity_factor
        return max(int(math.ceil(T / cfg.num_experts * cf * cfg.top_k_experts)), cfg.min_capacity)

    def _gate_draws(self, i, T, E, gumbel):
        """Random draws of the gate: an injected provider (tests: the same numbers go to the oracle) or, when the config asks for
        DeepSpeed's sampling behaviour, the stateless generator keyed by (seed, forward pass, layer)."""
        if self.rts_uniform_provider is not None:
            return self.rts_uniform_provider(i, T, E)
        if not self.cfg.moe_gate_sampling:
            return None
        # the generator is keyed by (seed, offset + index) and a pass's layers occupy consecutive offsets: ONE launch per forward pass draws
        # for all layers (the same numbers as a launch per layer, which sat between every layer's routing kernel and its expert GEMMs)
        L = len(self.layers)
        key = (self.gate_pass, T, E, bool(gumbel))
        if getattr(self, "_draws_key", None) != key:
            self._draws_all = ops.gate_noise(L * T * E, self.cfg.moe_gate_seed, self.gate_pass * L * T * E, gumbel, self.device)
            self._draws_key = key
        return self._draws_all[i * T * E:(i + 1) * T * E]

    def _mlp(self, i, lw, h, x, gate=None, needed=None, rstd=None):
        """x + MLP(h): h = post-attention RMSNorm output [T,d], x = residual stream [T,d]; gate = (logits, gates) when the caller's fused
        norm kernel already produced them (ops.rmsnorm_gate)."""
        cfg = self.cfg
        T = x.shape[0]
        if i not in self.moe_layers:
            if T <= 8:
                return ops.gemv(ops.gemv(h, lw["gu"], act=ops.ACT_SWIGLU_PAIR), lw["down"], residual=x), None, None
            act = ops.gemm(h, lw["gu"], act=ops.ACT_SWIGLU_PAIR)       # silu(gate)*up fused into the GEMM epilogue
            return ops.gemm(act, lw["down"], residual=x), None, None
        E, ff, d = cfg.num_experts, cfg.intermediate_size, cfg.hidden_size
        cap = self.capacity(T)
        k = cfg.top_k_experts
        logits, gates = gate if gate is not None else ops.moe_gate(h, lw["wg"])
        if k == 1 and self.ep is None and T <= 8 and not cfg.use_residual:
            # decode rows: each row streams its own expert's matrices (GEMV with a device-side expert index); the combine weight,
            # the capacity drop and the residual ride in the down projection's epilogue
            # (no gate draws: random token selection only acts when an expert is over capacity, and capacity >= T for these rows
            #  only when cap >= T; otherwise the draws are generated as usual)
            draws = None if cap >= T else self._gate_draws(i, T, E, gumbel=False)
            expert, slot, weight, kept, counts, l_aux = ops.moe_route_top1(gates, cap, draws)
            act = ops.gemv(h, lw["gu"], act=ops.ACT_SWIGLU_PAIR, w_index=expert)
            out = ops.gemv(act, lw["down"], residual=x, w_index=expert, row_scale=weight, row_keep=slot)
            return out, l_aux, (expert, slot, counts)
        if k == 1 and self.ep is None and self.fuse_moe_gather_scatter and not cfg.use_residual:
            # top-1 on one rank: the dispatch is a row gather in the gate|up GEMM's operand fetch and the combine (gate weight and
            # the layer's residual add) a row scatter in the down GEMM's epilogue; every routed token's row is written exactly once,
            # the capacity-dropped ones get the residual stream from a fill kernel
            expert, slot, weight, kept, counts, l_aux, slot_token = ops.moe_route_top1(
                gates, cap, self._gate_draws(i, T, E, gumbel=False), want_slot_token=True)
            act = torch.empty((E, cap, ff), dtype=torch.bfloat16, device=x.device)
            if needed is not None:
                self.pruned_rows = int(getattr(self, "needed_rows")[0].numel())      # the pruning HAPPENED (model_forward reports only this)
                # the last layer: only the rows something reads go through the experts (the routing above saw every token: capacity drops,
                # l_aux and the counts are those of the whole batch); a row that is not computed keeps the residual stream, which nobody reads
                slot_token, kept = ops.moe_filter_slots(slot_token, kept, needed)
            if ops.GEMM_TIMER is not None:
                ops.GEMM_TIMER.batched_tag = i          # the expert GEMMs are credited with the rows `kept` holds after the region
            ev = getattr(self, "layer_events", None)
            if ev is not None and i < len(ev):
                ev[i].record()          # "layer i's expert GEMMs start here": what the frozen towers of the NEXT step gate their layers on (medplib.model_forward)
            if rstd is not None:        # folded post-attention norm: the experts read the raw stream, W carries ln2, the epilogue applies rstd
                ops.gemm_batched_rows(x, lw["gu_f"], act, kept, a_rows=slot_token, act=ops.ACT_SWIGLU_PAIR, rows_stride=cap, a_row_scale=rstd)
            else:
                ops.gemm_batched_rows(h, lw["gu"], act, kept, a_rows=slot_token, act=ops.ACT_SWIGLU_PAIR, rows_stride=cap)
            # Round 4: the combine writes INTO the residual stream (out = residual = x).  Every (routed row, 16-byte column piece) has
            # exactly one owner in the down projection's epilogue — a tile, or one unit's share of a split tail tile — which reads the
            # residual piece and stores the sum at the same address; a capacity-dropped token's row is simply left as it is, which is
            # DeepSpeed's result for it (x + 0).  The moe_fill_dropped launch per layer and the [T, d] output buffer are gone (same bits:
            # test_moe_fused_gather_scatter_matches_unfused).  No gradient flows here (the LoRA path keeps its own forward).
            ops.gemm_batched_rows(act, lw["down"], x, kept, c_rows=slot_token, c_scale=weight, residual=x, rows_stride=cap)
            return x, l_aux, (expert, slot, counts)
        if k == 1:
            expert, slot, weight, kept, counts, l_aux = ops.moe_route_top1(gates, cap, self._gate_draws(i, T, E, gumbel=False))
        else:
            expert, slot, weight, kept, counts, l_aux = ops.moe_route_top2(gates, logits, cap, self._gate_draws(i, T, E, gumbel=True))
        if self.ep is not None:
            # slabs of capx + 1 rows: capx = the capacity agreed over the expert-parallel group for this pass (ranks see different
            # T), the extra row is the header that carries the row count through the same all-to-all (expert_parallel.py)
            capx = self.ep.exchange_capacity(cap, key=self.gate_pass)
            buf = ops.moe_dispatch(h, expert, slot, E, capx + 1, top_k=k)
            y = self._experts_parallel(lw, buf, kept, capx)
            if cfg.use_residual:
                raise NotImplementedError("residual MoE with ep_size > 1")
            return ops.moe_combine(y, expert, slot, weight, x, capx, top_k=k), l_aux, (expert, slot, counts)
        buf = ops.moe_dispatch(h, expert, slot, E, cap, top_k=k)
        act = torch.empty((E, cap, ff), dtype=torch.bfloat16, device=h.device)
        if ops.GEMM_TIMER is not None:
            ops.GEMM_TIMER.batched_tag = i              # (credited with the kept rows: every token visits k experts unless dropped)
        ops.gemm_batched(buf, lw["gu"], act, m_dev=kept, act=ops.ACT_SWIGLU_PAIR)
        y = torch.empty((E, cap, d), dtype=torch.bfloat16, device=h.device)
        ops.gemm_batched(act, lw["down"], y, m_dev=kept)
        if cfg.use_residual:
            # residual MoE: the routed output and a dense MLP of the same input, mixed by a learned two-way softmax
            moe = ops.moe_combine(y, expert, slot, weight, None, cap, top_k=k)
            mlp = ops.gemm(ops.gemm(h, lw["res_gu"], act=ops.ACT_SWIGLU_PAIR), lw["res_down"])
            coef = ops.gemm(h, lw["coef_w"], bias=lw["coef_b"])
            return ops.moe_residual_mix(x, moe, mlp, coef), l_aux, (expert, slot, counts)
        out = ops.moe_combine(y, expert, slot, weight, x, cap, top_k=k)
        return out, l_aux, (expert, slot, counts)

    def _experts_parallel(self, lw, buf, kept, cap):
        """MOELayer with ep_size > 1: all-to-all the routed rows to the ranks that own the experts, run each local expert over the
        [ep] capacity slabs it received (one batched GEMM pair per local expert, the slab row counts travel with the rows), and
        all-to-all the outputs back (sharded_moe.py MOELayer.forward; collectives C3 of SURVEY §2.5)."""
        cfg, ep = self.cfg, self.ep
        ff, d = cfg.intermediate_size, cfg.hidden_size
        recv, counts = ep.dispatch(buf, kept)                      # [ep, E_local, cap + 1, d] (row cap = header), [ep, E_local]
        counts_t = counts.t().contiguous()                         # [E_local, ep]: per local expert, rows from each source rank
        y = torch.empty((ep.ep, ep.E_local, cap, d), dtype=recv.dtype, device=recv.device)
        for e in range(ep.E_local):
            a = recv[:, e, :cap]                                   # [ep, cap, d] view, batch stride E_local*(cap+1)*d
            act = torch.empty((ep.ep, cap, ff), dtype=torch.bfloat16, device=buf.device)
            ops.gemm_batched(a, lw["gu"][e].unsqueeze(0).expand(ep.ep, -1, -1), act, m_dev=counts_t[e], act=ops.ACT_SWIGLU_PAIR)
            ops.gemm_batched(act, lw["down"][e].unsqueeze(0).expand(ep.ep, -1, -1), y[:, e], m_dev=counts_t[e])
        return ep.combine(y)

    def new_kv_cache(self, batch, max_len):
        """KV cache for `evaluate()`'s greedy decode (HF `use_cache=True`, MedPLIB.py:592-606): post-RoPE K and V per layer."""
        cfg = self.cfg
        shape = (batch, max_len, cfg.num_attention_heads, cfg.head_dim)
        return {"len": 0, "k": [torch.empty(shape, dtype=torch.bfloat16, device=self.device) for _ in self.layers],
                "v": [torch.empty(shape, dtype=torch.bfloat16, device=self.device) for _ in self.layers]}

    def forward(self, inputs_embeds, key_valid=None, collect_routing=False, kv_cache=None):
        """inputs_embeds [B,S,d] bf16; key_valid uint8 [B,S] (1 = real token) or None.
        With kv_cache: the S new tokens sit at positions cache.len .. cache.len+S-1, their K/V are appended and attention runs
        over the whole cache (prefill when cache.len == 0, single-token decode afterwards).
        Returns (last_hidden_state after the final RMSNorm [B,S,d], [l_aux per MoE layer], routing or None)."""
        cfg = self.cfg
        B, S, d = inputs_embeds.shape
        H, D = cfg.num_attention_heads, cfg.head_dim
        x = inputs_embeds.reshape(B * S, d)
        aux, routing, gate_inputs = [], [], []
        nr = getattr(self, "needed_rows", None)             # (rows, mask) from model_forward: the output rows something reads, or None
        needed_mask = nr[1] if (nr is not None and not collect_routing and kv_cache is None and nr[1].numel() == B * S) else None
        self.gate_pass += 1
        pos0 = kv_cache["len"] if kv_cache is not None else 0
        # a handful of rows (the single-token decode steps): the projections are weight streams -> GEMV kernel (HBM-bound)
        lin = (lambda a, w, **kw: ops.gemv(a, w, **kw)) if B * S <= 8 else (lambda a, w, **kw: ops.gemm(a, w, **kw))
        ff = cfg.intermediate_size
        # folded norms (config.fold_input_norm): frozen top-1 MoE layers of 320-row-kernel shapes, one rank, no intermediate capture
        fold = ("qkv_rope_f" in self.layers[0] if self.layers else False) and kv_cache is None and self.ep is None and self.fuse_moe_gather_scatter \
            and d in ops.RMSNORM_GATE_DIMS and ops.gemm_fold_ok(B * S, 3 * d, d) and ops.gemm_fold_ok(self.capacity(B * S), 2 * ff, d)
        self.folded_layers = 0
        for i, lw in enumerate(self.layers):
            if fold and "qkv_rope_f" in lw:
                # x -> rstd1; qkv = (x W'^T) * rstd1 -> RoPE; attention; o_proj + residual; x -> rstd2 + gate (from HF's bf16 h); experts on x with W''
                rstd1, _, _ = ops.rmsnorm_gate_rstd(x, lw["ln1"], cfg.rms_norm_eps)
                qkv = ops.gemm_qkv_rope(x, lw["qkv_rope_f"], self.cos, self.sin, S, H, D, pos_offset=pos0, out=ops.padded_rows(B * S, 3 * d, x.device),
                                        row_scale=rstd1)
                q5 = qkv.unflatten(0, (B, S)).unflatten(2, (3, H, D))
                attn = ops.attention(q5[:, :, 0], q5[:, :, 1], q5[:, :, 2], causal=True, key_valid=key_valid)
                x = lin(attn.view(B * S, d), lw["o"], residual=x)
                x_gate_in = x.clone() if collect_routing else None          # (tests / parity only, as below)
                rstd2, lg, gt = ops.rmsnorm_gate_rstd(x, lw["ln2"], cfg.rms_norm_eps, lw["wg"])
                x, l_aux, r = self._mlp(i, lw, None, x, gate=(lg, gt), needed=needed_mask if i == len(self.layers) - 1 else None, rstd=rstd2)
                aux.append(l_aux)
                if collect_routing:
                    routing.append(r)
                    gate_inputs.append(x_gate_in)
                self.folded_layers += 1
                continue
            h = ops.rmsnorm(x, lw["ln1"], cfg.rms_norm_eps)
            if self.fuse_rope and B * S > 8:
                # (row stride of the qkv buffer kept off multiples of 8 KiB: ops.padded_rows)
                qkv = ops.gemm_qkv_rope(h, lw["qkv_rope"], self.cos, self.sin, S, H, D, pos_offset=pos0,
                                        out=ops.padded_rows(B * S, 3 * d, x.device))                       # RoPE in the epilogue
            else:
                qkv = lin(h, lw["qkv"])
                ops.rope_qk_(qkv, self.cos, self.sin, S, H, D, pos_offset=pos0)
            q5 = qkv.unflatten(0, (B, S)).unflatten(2, (3, H, D))
            if kv_cache is not None:
                kv_cache["k"][i][:, pos0:pos0 + S].copy_(q5[:, :, 1])
                kv_cache["v"][i][:, pos0:pos0 + S].copy_(q5[:, :, 2])
            if kv_cache is not None and pos0 > 0:
                assert S == 1 and key_valid is None, "cached decode is single-token, un-padded (the reference evaluates with batch 1)"
                attn = ops.attention(q5[:, :, 0], kv_cache["k"][i][:, :pos0 + 1], kv_cache["v"][i][:, :pos0 + 1], causal=False)
            else:
                attn = ops.attention(q5[:, :, 0], q5[:, :, 1], q5[:, :, 2], causal=True, key_valid=key_valid)
            x = lin(attn.view(B * S, d), lw["o"], residual=x)
            # (tests / parity only: the layer's own gate input, copied before the combine writes the MLP output into the residual stream — the
            #  layer-local routing check of oracle/parity.py recomputes the gate from it)
            x_gate_in = x.clone() if (collect_routing and i in self.moe_layers) else None
            if i in self.moe_layers and d in ops.RMSNORM_GATE_DIMS and B * S > 8:
                # post-attention norm and the MoE gate in one pass over the rows (bit-identical with the two kernels)
                h, lg, gt = ops.rmsnorm_gate(x, lw["ln2"], cfg.rms_norm_eps, lw["wg"])
                x, l_aux, r = self._mlp(i, lw, h, x, gate=(lg, gt), needed=needed_mask if i == len(self.layers) - 1 else None)
            else:
                h = ops.rmsnorm(x, lw["ln2"], cfg.rms_norm_eps)
                # (the row set also reaches the MoE branch behind the unfused norm — hidden sizes without an rmsnorm_gate instantiation, the tiny
                #  test dims: _mlp prunes in its top-1 gather / scatter branch whichever kernel produced the gate)
                x, l_aux, r = self._mlp(i, lw, h, x, needed=needed_mask if (i in self.moe_layers and i == len(self.layers) - 1) else None)
            if l_aux is not None:
                aux.append(l_aux)
                if collect_routing:
                    routing.append(r)
                    gate_inputs.append(x_gate_in)
        if kv_cache is not None:
            kv_cache["len"] = pos0 + S
        out = ops.rmsnorm(x, self.norm_w, cfg.rms_norm_eps)
        if collect_routing:
            self.last_gate_inputs = gate_inputs            # per MoE layer: the residual stream in front of the post-attention norm (oracle/parity.py)
        return out.view(B, S, d), aux, (routing if collect_routing else None)

    def decode_step(self, emb, kv_cache, counters):
        """One token per sequence against the KV cache with the cache length held ON THE DEVICE (`counters` int32 [2] =
        [position of the new token, number of valid keys after appending it]): no launch argument depends on the step, so the
        whole step can be captured once into a HIP graph and replayed per token (evaluate()).  emb [B, 1, d] -> hidden [B, 1, d].
        The caller advances `counters` (ops.advance_ints) after the step.
        Reproducibility note (round-4 advisor): the narrow projections (o_proj, down) take the K-split GEMV when B == 1 or an expert index is
        given and the shared-weight form otherwise; the two add their partial dot products in different orders, so the SAME prompt decoded at
        batch 1 and at batch 2 may differ in the last fp32 bit of those projections (and, at a near-tie of two logits, in a greedy token).
        Within one batch size every step is bit-reproducible; MP_GEMV_KSPLIT=0 selects the shared form for every batch size."""
        cfg = self.cfg
        B, _, d = emb.shape
        H, D = cfg.num_attention_heads, cfg.head_dim
        x = emb.reshape(B, d)
        self.gate_pass += 1
        fold = ops.gemv_rmsnorm_ok(B, d, head_dim=D, swiglu_n=2 * cfg.intermediate_size)   # input_layernorm inside the qkv GEMV (same bits, one launch less per layer)
        for i, lw in enumerate(self.layers):
            if fold:                                            # norm + projection + RoPE + cache append: one launch, the three launches' bits
                qkv = ops.gemv_rmsnorm_rope_append(x, lw["ln1"], cfg.rms_norm_eps, lw["qkv"], self.cos, self.sin, kv_cache["k"][i],
                                                   kv_cache["v"][i], counters[0:1], H, D)
            else:
                qkv = ops.gemv(ops.rmsnorm(x, lw["ln1"], cfg.rms_norm_eps), lw["qkv"])
                ops.decode_rope_append(qkv, self.cos, self.sin, kv_cache["k"][i], kv_cache["v"][i], counters[0:1], H, D)
            q4 = qkv.view(B, 1, 3, H, D)[:, :, 0]
            attn = ops.attention(q4, kv_cache["k"][i], kv_cache["v"][i], causal=False, sk_dev=counters[1:2])
            x = ops.gemv(attn.view(B, d), lw["o"], residual=x)
            if (i in self.moe_layers and cfg.top_k_experts == 1 and self.ep is None and B <= 8 and self.fuse_decode_routing
                    and not cfg.use_residual):
                # post-attention norm + gate + routing in one launch, then the two expert GEMVs (same bits as the separate kernels)
                E, cap = cfg.num_experts, self.capacity(B)
                draws = None if cap >= B else self._gate_draws(i, B, E, gumbel=False)
                h, expert, slot, weight, _, _, _ = ops.decode_norm_gate_route(x, lw["ln2"], cfg.rms_norm_eps, lw["wg"], cap, draws)
                act = ops.gemv(h, lw["gu"], act=ops.ACT_SWIGLU_PAIR, w_index=expert)
                x = ops.gemv(act, lw["down"], residual=x, w_index=expert, row_scale=weight, row_keep=slot)
            elif fold and i not in self.moe_layers:
                # dense layer: post_attention_layernorm inside the gate|up GEMV (same bits as rmsnorm + gemv), then the down projection
                act = ops.gemv_rmsnorm(x, lw["ln2"], cfg.rms_norm_eps, lw["gu"], act=ops.ACT_SWIGLU_PAIR)
                x = ops.gemv(act, lw["down"], residual=x)
            else:
                h = ops.rmsnorm(x, lw["ln2"], cfg.rms_norm_eps)
                x, _, _ = self._mlp(i, lw, h, x)
        return ops.rmsnorm(x, self.norm_w, cfg.rms_norm_eps).view(B, 1, d)

    def next_token_logits(self, hidden_row):
        """fp32 logits of one position: lm_head(hidden).float() (medplib_moe_llama.py:388-389).  hidden_row [n, d] bf16."""
        h = hidden_row.contiguous()
        if h.shape[0] <= 8:
            return ops.gemv(h, self.lm_head, out_dtype=torch.float32)
        return ops.gemm(h, self.lm_head, out_dtype=torch.float32)

    def cross_entropy(self, last_hidden, sup_rows, sup_labels, aux):
        """CE over supervised rows only (row-wise op; unsupervised rows never reach the loss — SURVEY B.8):
        fp32 logits = lm_head(hidden[sup_rows]) (medplib_moe_llama.py:388-389), mean CE (:392-408),
        + router_aux_loss_coef * sum(l_aux) (:410-421).  sup_rows int64 [n] flat row ids, sup_labels int64 [n]."""
        cfg = self.cfg
        d = cfg.hidden_size
        if sup_rows.numel() == 0:
            return torch.full((1,), float("nan"), dtype=torch.float32, device=last_hidden.device)
        rows = ops.gather_rows_bf16_to_f32(last_hidden.view(-1, d), sup_rows)
        rows = ops.cast_to_bf16(rows)                     # exact: the rows were bf16
        logits = ops.gemm(rows, self.lm_head, out_dtype=torch.float32)
        row_loss = ops.cross_entropy_rows(logits, sup_labels)
        add = torch.cat(aux) if (aux and cfg.router_aux_loss_coef != 0.0) else None
        return ops.mean_plus(row_loss, 1.0, add, cfg.router_aux_loss_coef)
