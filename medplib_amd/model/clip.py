"""CLIP ViT-L/14-336 vision tower + mm_projector over the HIP kernels.

Mirrors `CLIPVisionTower.forward` + `feature_select` (model/medplib/model/multimodal_encoder/clip_encoder.py:31-60: frozen,
hidden_states[mm_vision_select_layer], drop CLS) and HF-4.31 CLIPVisionModel arithmetic, then
`mm_projector` mlp2x_gelu (multimodal_projector/builder.py:39-46).  Only the layers hidden_states[select_layer]
depends on are run (the reference also runs the unused last layer)."""
import torch

from .. import ops


class ClipTower:
    def __init__(self, cfg, device, seed=1, init_std=0.02):
        self.cfg, self.device = cfg, device
        C, I, L = cfg.clip_hidden_size, cfg.clip_intermediate_size, cfg.clip_num_layers
        p = cfg.clip_patch_size
        self.k_patch = 3 * p * p
        self.k_pad = -(-self.k_patch // 64) * 64
        g = torch.Generator(device=device).manual_seed(seed)

        def rn(*shape, std=init_std, dtype=torch.bfloat16):
            return (torch.randn(*shape, generator=g, device=device, dtype=torch.float32) * std).to(dtype)

        def ones(n):
            return torch.ones(n, dtype=torch.float32, device=device)

        def zeros(n):
            return torch.zeros(n, dtype=torch.float32, device=device)
        self.patch_w = torch.zeros(C, self.k_pad, dtype=torch.bfloat16, device=device)
        self.patch_w[:, :self.k_patch] = rn(C, self.k_patch)
        self.cls = rn(C)
        self.pos = rn(cfg.clip_num_patches + 1, C)
        self.pre_ln = (ones(C), zeros(C))
        self.layers = []
        for _ in range(L):
            self.layers.append({"ln1": (ones(C), zeros(C)), "ln2": (ones(C), zeros(C)),
                                "qkv_w": rn(3 * C, C), "qkv_b": rn(3 * C, dtype=torch.float32),
                                "o_w": rn(C, C), "o_b": rn(C, dtype=torch.float32),
                                "fc1_w": rn(I, C), "fc1_b": rn(I, dtype=torch.float32),
                                "fc2_w": rn(C, I), "fc2_b": rn(C, dtype=torch.float32)})
        d = cfg.hidden_size
        self.proj = {"w0": rn(d, C), "b0": rn(d, dtype=torch.float32), "w2": rn(d, d), "b2": rn(d, dtype=torch.float32)}
        # region_fea_adapter = nn.Linear(mm_hidden_size, hidden_size) on the RAW tower features (medplib_arch.py:131, 204-208)
        self.region_adapter = {"w": rn(d, C), "b": rn(d, dtype=torch.float32)}

    def n_run_layers(self):
        sel, L = self.cfg.mm_vision_select_layer, self.cfg.clip_num_layers
        idx = sel if sel >= 0 else L + 1 + sel      # hidden_states has L+1 entries; [k] = input of layer k
        return idx

    # ------------------------------------------------------------------ HF checkpoint layout
    def load_hf(self, sd, tower_prefix="model.vision_tower.vision_tower.vision_model.", proj_prefix="model.mm_projector."):
        C = self.cfg.clip_hidden_size

        def put(dst, src):
            dst.copy_(src.to(device=dst.device, dtype=dst.dtype))
        tp = tower_prefix
        self.patch_w.zero_()
        put(self.patch_w[:, :self.k_patch], sd[tp + "embeddings.patch_embedding.weight"].reshape(C, -1))
        put(self.cls, sd[tp + "embeddings.class_embedding"])
        put(self.pos, sd[tp + "embeddings.position_embedding.weight"])
        put(self.pre_ln[0], sd[tp + "pre_layrnorm.weight"]); put(self.pre_ln[1], sd[tp + "pre_layrnorm.bias"])
        for i, lw in enumerate(self.layers):
            lp = f"{tp}encoder.layers.{i}."
            for n, k in (("layer_norm1", "ln1"), ("layer_norm2", "ln2")):
                put(lw[k][0], sd[lp + n + ".weight"]); put(lw[k][1], sd[lp + n + ".bias"])
            for j, n in enumerate(("q", "k", "v")):
                put(lw["qkv_w"][j * C:(j + 1) * C], sd[lp + f"self_attn.{n}_proj.weight"])
                put(lw["qkv_b"][j * C:(j + 1) * C], sd[lp + f"self_attn.{n}_proj.bias"])
            put(lw["o_w"], sd[lp + "self_attn.out_proj.weight"]); put(lw["o_b"], sd[lp + "self_attn.out_proj.bias"])
            put(lw["fc1_w"], sd[lp + "mlp.fc1.weight"]); put(lw["fc1_b"], sd[lp + "mlp.fc1.bias"])
            put(lw["fc2_w"], sd[lp + "mlp.fc2.weight"]); put(lw["fc2_b"], sd[lp + "mlp.fc2.bias"])
        put(self.proj["w0"], sd[proj_prefix + "0.weight"]); put(self.proj["b0"], sd[proj_prefix + "0.bias"])
        put(self.proj["w2"], sd[proj_prefix + "2.weight"]); put(self.proj["b2"], sd[proj_prefix + "2.bias"])
        if "model.region_fea_adapter.weight" in sd:
            put(self.region_adapter["w"], sd["model.region_fea_adapter.weight"]); put(self.region_adapter["b"], sd["model.region_fea_adapter.bias"])

    def export_hf(self, tower_prefix="model.vision_tower.vision_tower.vision_model.", proj_prefix="model.mm_projector."):
        cfg = self.cfg
        C, p = cfg.clip_hidden_size, cfg.clip_patch_size
        bf = torch.bfloat16
        tp = tower_prefix
        sd = {tp + "embeddings.patch_embedding.weight": self.patch_w[:, :self.k_patch].reshape(C, 3, p, p),
              tp + "embeddings.class_embedding": self.cls, tp + "embeddings.position_embedding.weight": self.pos,
              tp + "pre_layrnorm.weight": self.pre_ln[0].to(bf), tp + "pre_layrnorm.bias": self.pre_ln[1].to(bf)}
        for i, lw in enumerate(self.layers):
            lp = f"{tp}encoder.layers.{i}."
            for n, k in (("layer_norm1", "ln1"), ("layer_norm2", "ln2")):
                sd[lp + n + ".weight"] = lw[k][0].to(bf); sd[lp + n + ".bias"] = lw[k][1].to(bf)
            for j, n in enumerate(("q", "k", "v")):
                sd[lp + f"self_attn.{n}_proj.weight"] = lw["qkv_w"][j * C:(j + 1) * C]
                sd[lp + f"self_attn.{n}_proj.bias"] = lw["qkv_b"][j * C:(j + 1) * C].to(bf)
            sd[lp + "self_attn.out_proj.weight"] = lw["o_w"]; sd[lp + "self_attn.out_proj.bias"] = lw["o_b"].to(bf)
            sd[lp + "mlp.fc1.weight"] = lw["fc1_w"]; sd[lp + "mlp.fc1.bias"] = lw["fc1_b"].to(bf)
            sd[lp + "mlp.fc2.weight"] = lw["fc2_w"]; sd[lp + "mlp.fc2.bias"] = lw["fc2_b"].to(bf)
        sd[proj_prefix + "0.weight"] = self.proj["w0"]; sd[proj_prefix + "0.bias"] = self.proj["b0"].to(bf)
        sd[proj_prefix + "2.weight"] = self.proj["w2"]; sd[proj_prefix + "2.bias"] = self.proj["b2"].to(bf)
        sd["model.region_fea_adapter.weight"] = self.region_adapter["w"]; sd["model.region_fea_adapter.bias"] = self.region_adapter["b"].to(bf)
        return sd

    # ------------------------------------------------------------------ forward
    @ops.with_throughput_tiles
    def encode_images(self, images, return_raw=False):
        """images [n,3,336,336] (bf16 or f32) -> projected features [n*576, hidden] bf16 (encode_images,
        medplib_arch.py:198-212, without compressor); with return_raw also the tower features [n*576, C] the region adapter
        reads."""
        cfg = self.cfg
        n = images.shape[0]
        C, NP, H = cfg.clip_hidden_size, cfg.clip_num_patches, cfg.clip_num_heads
        S = NP + 1
        cols = ops.patch_im2col(images.contiguous(), cfg.clip_patch_size, self.k_pad)
        patches = ops.gemm(cols, self.patch_w)
        x = ops.clip_embed(patches, self.cls, self.pos, n, NP, C).view(n * S, C)
        x = ops.layernorm(x, self.pre_ln[0], self.pre_ln[1], cfg.clip_ln_eps)
        gate = getattr(self, "gate_events", None)           # set by model_forward for a tower that runs ahead (see there): layer j waits for events[j]
        for j, lw in enumerate(self.layers[: self.n_run_layers()]):
            if gate is not None and j < len(gate):
                torch.cuda.current_stream().wait_event(gate[j])
            h = ops.layernorm(x, lw["ln1"][0], lw["ln1"][1], cfg.clip_ln_eps)
            qkv = ops.gemm(h, lw["qkv_w"], bias=lw["qkv_b"])
            q5 = qkv.view(n, S, 3, H, C // H)
            a = ops.attention(q5[:, :, 0], q5[:, :, 1], q5[:, :, 2])
            x = ops.gemm(a.view(n * S, C), lw["o_w"], bias=lw["o_b"], residual=x)
            h = ops.layernorm(x, lw["ln2"][0], lw["ln2"][1], cfg.clip_ln_eps)
            h = ops.gemm(h, lw["fc1_w"], bias=lw["fc1_b"], act=ops.ACT_QUICK_GELU)
            x = ops.gemm(h, lw["fc2_w"], bias=lw["fc2_b"], residual=x)
        feats = ops.copy_rows(x, n * NP, C, NP, S, 1)            # drop CLS (clip_encoder.py:33-34)
        h = ops.gemm(feats, self.proj["w0"], bias=self.proj["b0"], act=ops.ACT_GELU)
        out = ops.gemm(h, self.proj["w2"], bias=self.proj["b2"])
        return (out, feats) if return_raw else out

    def region_feature_map(self, raw_rows):
        """region_fea_adapter over raw tower features [rows, C] -> [rows, hidden] (medplib_arch.py:207)."""
        return ops.gemm(raw_rows, self.region_adapter["w"], bias=self.region_adapter["b"])
