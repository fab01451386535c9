"""Generates tests/golden/*.npz by running the REFERENCE (imported read-only from /root/reference, dev container only)
next to the oracle restatement on seeded inputs/weights.  The fixtures are data (inputs + expected outputs); no
reference source travels.  Run:  python -m oracle.make_golden   (needs /root/reference)."""
import os
import sys

import numpy as np
import torch

from . import ops, sam

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def _ref_sam():
    sam_dir = os.path.join(REF, "model", "segment_anything_med2d")
    sys.path.insert(0, sam_dir)
    try:
        import modeling  # noqa: the reference's own package (torch-only)
    finally:
        # the SAM-Med2D directory has a `utils` package of its own: left on sys.path (or in sys.modules) it shadows the reference's root
        # `utils` when a later target of the same process imports model.MedPLIB (round-4 review: the documented default run failed there)
        sys.path.remove(sam_dir)
        for name in [k for k, v in sys.modules.items() if (k == "utils" or k.startswith("utils."))
                     and os.path.abspath(getattr(v, "__file__", "") or "").startswith(sam_dir)]:
            del sys.modules[name]
    from functools import partial
    enc = modeling.ImageEncoderViT(depth=12, embed_dim=768, img_size=256, mlp_ratio=4,
                                   norm_layer=partial(torch.nn.LayerNorm, eps=1e-6), num_heads=12, patch_size=16,
                                   qkv_bias=True, use_rel_pos=True, global_attn_indexes=[2, 5, 8, 11], window_size=14,
                                   out_chans=256, adapter_train=True)
    pe = modeling.PromptEncoder(embed_dim=256, image_embedding_size=(16, 16), input_image_size=(256, 256), mask_in_chans=16)
    dec = modeling.MaskDecoder(num_multimask_outputs=3,
                               transformer=modeling.TwoWayTransformer(depth=2, embedding_dim=256, mlp_dim=2048, num_heads=8),
                               transformer_dim=256, iou_head_depth=3, iou_head_hidden_dim=256)
    return enc, pe, dec


def _load(mod, W, prefix):
    sd = {k[len(prefix) + 1:]: v for k, v in W.items() if k.startswith(prefix + ".")}
    missing, unexpected = mod.load_state_dict(sd, strict=False)
    # keys the oracle's weight dict does not carry are prompt-encoder parts the text-only path never touches
    assert not unexpected, unexpected
    return missing


def golden_sam():
    torch.manual_seed(0)
    W = sam.init_weights(seed=1234)
    enc, pe, dec = _ref_sam()
    m1 = _load(enc, W, "image_encoder"); assert not m1, m1
    m2 = _load(pe, W, "prompt_encoder")
    m3 = _load(dec, W, "mask_decoder"); assert not m3, m3
    enc.eval(); pe.eval(); dec.eval()
    g = torch.Generator().manual_seed(42)
    img = torch.randn(1, 3, 256, 256, generator=g)
    with torch.no_grad():
        ref_emb = enc(img)
        ora_emb = sam.image_encoder(img, W)
    d = (ref_emb - ora_emb).abs().max().item()
    print("image_encoder max|ref-oracle| =", d, "ref absmax", ref_emb.abs().max().item())
    assert d < 2e-4

    text = torch.randn(2, 1, 256, generator=g) * 0.5
    emb2 = torch.cat([ref_emb, ref_emb.flip(-1)], 0)
    with torch.no_grad():
        r_masks, r_iou = [], []
        for i in range(2):   # the reference runs one prompt at a time (model/MedPLIB.py:473-502)
            sp, de = pe(points=None, boxes=None, masks=None, text_embeds=text[i:i + 1])
            lm, io = dec(image_embeddings=emb2[i:i + 1], image_pe=pe.get_dense_pe(), sparse_prompt_embeddings=sp,
                         dense_prompt_embeddings=de, multimask_output=False)
            r_masks.append(lm); r_iou.append(io)
        r_masks, r_iou = torch.cat(r_masks), torch.cat(r_iou)
        r_pe = pe.get_dense_pe()
        o_pe = sam.dense_pe(W)
        sp, de = sam.prompt_encoder_text(text, W)
        o_masks, o_iou = sam.mask_decoder(emb2, o_pe, sp, de, W)
    print("dense_pe diff", (r_pe - o_pe).abs().max().item())
    print("mask_decoder masks diff", (r_masks - o_masks).abs().max().item(), "iou diff", (r_iou - o_iou).abs().max().item(),
          "mask absmax", r_masks.abs().max().item())
    assert torch.equal(r_pe, o_pe)
    assert (r_masks - o_masks).abs().max().item() < 1e-4 and (r_iou - o_iou).abs().max().item() < 1e-5
    np.savez_compressed(os.path.join(OUT, "sam_reference.npz"), weight_seed=np.int64(1234), image=img.numpy(),
                        image_embedding=ref_emb.numpy(), text_embeds=text.numpy(), dense_pe=r_pe.numpy(),
                        low_res_masks=r_masks.numpy(), iou_pred=r_iou.numpy())


def _import_reference_medplib():
    """Import the reference's model/MedPLIB.py.  It pulls deepspeed / torchvision / cv2 at import time, none of which exist in
    the image, so those names are registered as EMPTY types.ModuleType stand-ins (no behaviour is faked: nothing on the paths
    exercised here calls into them)."""
    import types
    import transformers  # noqa: F401  (must be fully imported before the stubs go in: its lazy loader probes find_spec)
    import transformers.modeling_utils, transformers.generation  # noqa: F401,E401
    import transformers.models.llama.modeling_llama, transformers.models.clip.modeling_clip  # noqa: F401,E401
    from transformers import (AutoConfig, AutoModelForCausalLM, BitsAndBytesConfig, CLIPImageProcessor,  # noqa: F401
                              CLIPVisionConfig, CLIPVisionModel, LlamaConfig, LlamaForCausalLM, LlamaModel)
    for name in ["deepspeed", "deepspeed.moe", "deepspeed.moe.layer", "torchvision", "torchvision.transforms",
                 "torchvision.transforms.functional", "torchvision.ops", "torchvision.ops.boxes", "cv2"]:
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)
    sys.modules["deepspeed.moe.layer"].MoE = type("MoE", (torch.nn.Module,), {})
    sys.modules["torchvision.transforms.functional"].resize = lambda *a, **k: None
    sys.modules["torchvision.transforms.functional"].to_pil_image = lambda *a, **k: None
    sys.modules["torchvision.ops.boxes"].batched_nms = lambda *a, **k: None
    sys.modules["torchvision.ops.boxes"].box_area = lambda *a, **k: None
    if REF in sys.path:
        sys.path.remove(REF)
    sys.path.insert(0, REF)                                # first: `utils`, `model`, `datasets` must resolve to the reference's root packages
    try:
        import model.MedPLIB as M
    except Exception as e:  # pragma: no cover
        print("reference model.MedPLIB import failed:", repr(e))
        raise
    return M


def golden_glue():
    """Splice / <SEG> mask / TokenCompressor / MaskTokenEncoder: run the reference's own
    `LlavaMetaForCausalLM.prepare_inputs_labels_for_multimodal` (medplib_arch.py:217-527), `build_seg_token_mask`
    (MedPLIB.py:310-355), `TokenCompressor` / `MaskTokenEncoder` (medplib_arch.py:67-108) on seeded tiny-dim inputs.  The mixin
    is hosted by a minimal nn.Module that supplies what it asks of `get_model()` (embed_tokens, mm_projector, a vision tower
    that returns a fixed linear read-out of the pixels, compressor, mask encoder)."""
    import types
    from . import llm
    M = _import_reference_medplib()
    import model.medplib.model.medplib_arch as A
    d, Cv, P, V = 8, 4, 6, 40          # hidden, vision width, patches per image, vocab
    SEG = 33
    g = torch.Generator().manual_seed(11)

    class Tower(torch.nn.Module):
        hidden_size = Cv

        def __init__(self, n_patches=P):
            super().__init__()
            self.num_patches = n_patches

        def forward(self, images):
            n = self.num_patches
            return images.flatten(1)[:, :n * Cv].reshape(images.shape[0], n, Cv)

        @property
        def dummy_feature(self):
            return torch.zeros(1, Cv)

    class Inner(torch.nn.Module):
        def __init__(self, compress, mask_enc, n_patches=P):
            super().__init__()
            self.embed_tokens = torch.nn.Embedding(V, d)
            self.mm_projector = torch.nn.Linear(Cv, d)
            self.vision_tower = Tower(n_patches)
            self.max_sample_point = 20
            self.region_fea_adapter = torch.nn.Linear(Cv, d)
            if compress:
                self.mm_token_compressor = A.TokenCompressor(d, compress)
            if mask_enc:
                self.mask_encoder = A.MaskTokenEncoder(d, mask_enc)

        def get_vision_tower(self):
            return self.vision_tower

    class Host(torch.nn.Module, A.LlavaMetaForCausalLM):
        def __init__(self, compress=0, mask_enc=0, n_patches=P):
            torch.nn.Module.__init__(self)
            self.inner = Inner(compress, mask_enc, n_patches)
            self.config = types.SimpleNamespace(mm_use_im_start_end=True, tune_mm_mlp_adapter=False, mm_token_compress=bool(compress),
                                                mm_compressed_token_count=compress, icl_mask_encoder=bool(mask_enc),
                                                region_geo_sampler=False)
            self.seg_token_idx = SEG
            self.device = torch.device("cpu")

        def get_model(self):
            return self.inner

        def get_input_embeddings(self):
            return self.inner.embed_tokens

        def get_output_embeddings(self):
            return self.inner.mm_projector

    def seed_params(mod, seed):
        gg = torch.Generator().manual_seed(seed)
        with torch.no_grad():
            for prm in mod.parameters():
                prm.copy_(torch.randn(prm.shape, generator=gg) * 0.3)

    def weights_of(host):
        return {"model." + k: v.detach().clone() for k, v in host.inner.state_dict().items()}

    out = {}

    def ids_row(L, img_pos, seg_pos, pad_from=None):
        r = torch.randint(3, 30, (L,), generator=g)
        for p_ in img_pos:
            r[p_] = -200
            r[p_ - 1] = 31; r[p_ + 1] = 32            # <im_start>, <im_end>
        for p_ in seg_pos:
            r[p_] = SEG
        if pad_from is not None:
            r[pad_from:] = 0
        return r

    def run_case(tag, host, ids, labels, att, images, mask_images=None, types_=None, lengths=None, default_len=None):
        host.eval()
        with torch.no_grad():
            _, new_att, _, emb, new_lab = host.prepare_inputs_labels_for_multimodal(ids, att, None, labels, images, None, None,
                                                                                     mask_images=mask_images, image_token_types=types_)
            seg = M.MedPLIBForCausalLM.build_seg_token_mask(host, ids, image_token_len=default_len, image_token_lengths=lengths)
        # ---- the restatement must agree bit for bit
        W = weights_of(host)
        with torch.no_grad():
            if isinstance(images, (list, tuple)) or images.dim() == 5:
                cat = torch.cat([im for im in images], 0)
            else:
                cat = images
            # the host's projector is a single Linear (the mlp2x projector itself is pinned through clip/projector tests)
            feats = torch.nn.functional.linear(host.inner.vision_tower(cat), W["model.mm_projector.weight"], W["model.mm_projector.bias"])
            if host.config.mm_token_compress:
                feats = llm.token_compressor(feats, W, host.config.mm_compressed_token_count)
            per_token = False
            if types_ is not None and mask_images is not None and len(mask_images) > 0:
                mf = llm.mask_token_encoder(torch.cat([m for m in mask_images], 0), W, host.inner.mask_encoder.num_tokens)
                feat_list = llm.combine_icl_features(list(feats), list(mf), types_)
                per_token = True
            elif isinstance(images, (list, tuple)) or images.dim() == 5:
                feat_list = list(feats); per_token = True
            else:
                feat_list = feats
            o_att, o_emb, o_lab = llm.prepare_inputs_labels_for_multimodal(ids, att, labels, feat_list, W["model.embed_tokens.weight"], per_token)
            o_seg = llm.build_seg_token_mask(ids, SEG, default_len, lengths)
        assert torch.equal(o_lab, new_lab) and torch.equal(o_att, new_att) and torch.equal(o_seg, seg), tag
        assert torch.allclose(o_emb, emb, rtol=0, atol=1e-6), (tag, (o_emb - emb).abs().max())
        out[f"{tag}_ids"] = ids.numpy(); out[f"{tag}_labels"] = labels.numpy(); out[f"{tag}_att"] = att.numpy()
        out[f"{tag}_new_labels"] = new_lab.numpy(); out[f"{tag}_new_att"] = new_att.numpy(); out[f"{tag}_embeds"] = emb.numpy()
        out[f"{tag}_seg_mask"] = seg.numpy()
        for k, v in W.items():
            if "mask_encoder" not in k:          # regenerated from llm.init_icl_weights(d, icl_seed) by the tests
                out[f"{tag}_W_{k}"] = v.numpy()
        if isinstance(images, (list, tuple)):
            out[f"{tag}_n_images"] = np.array([im.shape[0] for im in images]); out[f"{tag}_images"] = torch.cat(list(images), 0).numpy()
        else:
            out[f"{tag}_images"] = images.numpy()
        if mask_images is not None:
            out[f"{tag}_n_masks"] = np.array([m.shape[0] for m in mask_images]); out[f"{tag}_mask_images"] = torch.cat(list(mask_images), 0).numpy()
        if types_ is not None:
            out[f"{tag}_types"] = np.array([[1 if t == "mask" else 0 for t in row] for row in types_], dtype=np.int64)
        if lengths is not None:
            out[f"{tag}_lengths"] = np.array(lengths, dtype=np.int64)
        print("glue case", tag, "S =", emb.shape[1], "ok")

    # ---- case A: one image per sample (4-D images), a sample WITHOUT a placeholder, ragged right padding, two <SEG>
    host = Host(); seed_params(host, 1)
    L = 14
    ids = torch.stack([ids_row(L, [4], [9, 12]), ids_row(L, [], [6]), ids_row(L, [7], [11], pad_from=12)])
    att = torch.ones(3, L, dtype=torch.bool); att[2, 12:] = False
    labels = ids.clone(); labels[:, :5] = -100; labels[2, 12:] = -100
    images = torch.randn(3, 3, 4, 4, generator=g)
    run_case("A", host, ids, labels, att, images, default_len=P)

    # ---- case B: multi-image list layout (ICL overlay: 3 and 2 images), compressor 6 -> 4 tokens
    host = Host(compress=4); seed_params(host, 2)
    L = 20
    ids = torch.stack([ids_row(L, [3, 8, 13], [17]), ids_row(L, [5, 11], [15], pad_from=18)])
    att = torch.ones(2, L, dtype=torch.bool); att[1, 18:] = False
    labels = ids.clone(); labels[:, :14] = -100
    images = [torch.randn(3, 3, 4, 4, generator=g), torch.randn(2, 3, 4, 4, generator=g)]
    run_case("B", host, ids, labels, att, images, default_len=4)

    # ---- case C: ICL separate mode with mask encoder: [image, mask] x 2 + [image]; mask encoder 3 tokens; compressor 4
    host = Host(compress=4, mask_enc=3); seed_params(host, 3)
    Wc = llm.init_icl_weights(d, 303)
    host.inner.mask_encoder.load_state_dict({k[len("model.mask_encoder."):]: v for k, v in Wc.items() if k.startswith("model.mask_encoder.")})
    out["C_icl_seed"] = np.int64(303)
    L = 26
    ids = torch.stack([ids_row(L, [2, 6, 10, 14, 18], [23]), ids_row(L, [3, 7, 11, 15, 19], [22, 24])])
    att = torch.ones(2, L, dtype=torch.bool)
    labels = ids.clone(); labels[:, :20] = -100
    images = [torch.randn(3, 3, 4, 4, generator=g), torch.randn(3, 3, 4, 4, generator=g)]
    mask_images = [(torch.rand(2, 1, 32, 32, generator=g) > 0.6).float(), (torch.rand(2, 1, 32, 32, generator=g) > 0.4).float()]
    types_ = [["image", "mask", "image", "mask", "image"]] * 2
    lengths = [[4, 3, 4, 3, 4]] * 2
    run_case("C", host, ids, labels, att, images, mask_images, types_, lengths, default_len=4)

    # ---- case D: region prompts (medplib_arch.py:283-295, 409-433, 580-613): 9 patches (3 x 3 feature map), region tokens after
    #      the image; sample 1 has no region; one mask has more non-zero pixels than max_sample_point (20) -> torch.randperm
    host = Host(n_patches=9); seed_params(host, 4)
    L = 18
    ids = torch.stack([ids_row(L, [3], [15]), ids_row(L, [3], [14]), ids_row(L, [5], [16])])
    ids[0, 8] = -300; ids[0, 11] = -300; ids[2, 9] = -300
    att = torch.ones(3, L, dtype=torch.bool)
    labels = ids.clone(); labels[:, :12] = -100
    images = torch.randn(3, 3, 4, 4, generator=g)
    def blob(h, w, n):
        m = torch.zeros(h, w)
        idx = torch.randperm(h * w, generator=g)[:n]
        m.view(-1)[idx] = 1.0
        return m
    region_masks = [[blob(12, 10, 7), blob(12, 10, 45)], [blob(12, 10, 13)]]
    vrb = [[True, True], [False], [True]]
    host.eval()
    torch.manual_seed(123)
    with torch.no_grad():
        _, new_att, _, emb, new_lab = host.prepare_inputs_labels_for_multimodal(ids, att, None, labels, images, region_masks, vrb)
    Wd = weights_of(host)
    with torch.no_grad():
        raw = host.inner.vision_tower(images)
        feats = torch.nn.functional.linear(raw, Wd["model.mm_projector.weight"], Wd["model.mm_projector.bias"])
        rmap = torch.nn.functional.linear(raw, Wd["model.region_fea_adapter.weight"], Wd["model.region_fea_adapter.bias"])
        valid = torch.tensor([any(v) for v in vrb])
        torch.manual_seed(123)
        rfeat = llm.extract_region_feature(rmap[valid], region_masks, 20)
        o_att, o_emb, o_lab = llm.prepare_inputs_labels_for_multimodal(ids, att, labels, feats, Wd["model.embed_tokens.weight"], False,
                                                                       region_features=rfeat, valid_region_masks_bool=valid)
    assert torch.equal(o_lab, new_lab) and torch.equal(o_att, new_att)
    assert torch.allclose(o_emb, emb, rtol=0, atol=1e-6), (o_emb - emb).abs().max()
    out.update(D_ids=ids.numpy(), D_labels=labels.numpy(), D_att=att.numpy(), D_images=images.numpy(), D_new_labels=new_lab.numpy(),
               D_new_att=new_att.numpy(), D_embeds=emb.numpy(), D_valid=valid.numpy(), D_seed=np.int64(123), D_max_sample_point=np.int64(20),
               D_region_masks=np.stack([m.numpy() for ms in region_masks for m in ms]), D_region_counts=np.array([len(ms) for ms in region_masks]),
               D_region_features=torch.cat(rfeat).numpy())
    for k, v in Wd.items():
        out[f"D_W_{k}"] = v.numpy()
    print("glue case D (region prompts) S =", emb.shape[1], "ok")

    # ---- the two modules at the real token counts (576 -> 256; 336x336 mask -> 441 -> 64); weights come from
    #      llm.init_icl_weights(seed) on both sides, so only inputs and expected outputs are stored
    hid, wseed = 64, 77
    Wi = llm.init_icl_weights(hid, wseed)
    tc = A.TokenCompressor(hid, 256)
    tc.load_state_dict({k[len("model.mm_token_compressor."):]: v for k, v in Wi.items() if k.startswith("model.mm_token_compressor.")})
    x = torch.randn(1, 576, hid, generator=g)
    with torch.no_grad():
        y = tc(x)
    assert torch.equal(llm.token_compressor(x, Wi, 256), y)
    me = A.MaskTokenEncoder(hid, 64)
    me.load_state_dict({k[len("model.mask_encoder."):]: v for k, v in Wi.items() if k.startswith("model.mask_encoder.")})
    mk = (torch.rand(2, 1, 336, 336, generator=g) > 0.5).float()
    with torch.no_grad():
        ym = me(mk)
    assert torch.allclose(llm.mask_token_encoder(mk, Wi, 64), ym, rtol=0, atol=1e-6)
    out.update(icl_hidden=np.int64(hid), icl_weight_seed=np.int64(wseed), tc_x=x.numpy(), tc_y=y.numpy(),
               me_mask_bits=np.packbits(mk.numpy().astype(np.uint8)), me_y=ym.numpy(),
               icl_weight_checksum=np.float64(sum(float(v.double().sum()) for v in Wi.values())))
    np.savez_compressed(os.path.join(OUT, "glue_reference.npz"), **out)
    print("glue goldens ok")


def check_collate_contract():
    """Not a fixture: runs the reference's own DataCollatorForSupervisedDataset (datasets/DataCollatorForSupervisedDataset.py) on
    seeded per-sample dicts next to medplib_amd.collate.collate and requires identical batches key by key (tensors bit-equal,
    lists equal).  The CPU test tests/test_host_logic.py::test_collate_contract checks the same cases against expectations
    derived here (printed as literals)."""
    import importlib.util
    import types
    _import_reference_medplib()                        # puts REF on sys.path with the empty stand-ins registered
    spec = importlib.util.spec_from_file_location("ref_collator", os.path.join(REF, "datasets", "DataCollatorForSupervisedDataset.py"))
    mod = importlib.util.module_from_spec(spec); spec.loader.exec_module(mod)
    from medplib_amd.collate import collate
    sys.path.insert(0, os.path.join(os.path.dirname(OUT)))          # tests/ (not a package)
    from collate_cases import make_cases
    for name, samples in make_cases().items():
        for inf in (False, True):
            a = mod.DataCollatorForSupervisedDataset(samples, inference=inf)
            b = collate(samples, inference=inf)
            assert set(a) == set(b), (name, set(a) ^ set(b))
            for k in a:
                _same(a[k], b[k], f"{name}.{k}")
        print("collate case", name, "identical to the reference collator; input_ids", tuple(a["input_ids"].shape))


def _same(x, y, tag):
    if torch.is_tensor(x):
        assert torch.is_tensor(y) and x.shape == y.shape and x.dtype == y.dtype and torch.equal(x, y), tag
    elif isinstance(x, (list, tuple)):
        assert isinstance(y, (list, tuple)) and len(x) == len(y), tag
        for i, (p, q) in enumerate(zip(x, y)):
            _same(p, q, f"{tag}[{i}]")
    else:
        assert x == y, (tag, x, y)


def golden_mask_head():
    """postprocess_masks / losses / metrics: run the reference functions from model/MedPLIB.py.  That module imports
    deepspeed/transformers at import time, so the four loss callables and postprocess_masks are exercised through a
    minimal stub environment (types.ModuleType stand-ins for deepspeed & friends)."""
    M = _import_reference_medplib()
    g = torch.Generator().manual_seed(7)
    cases = []
    post = M.MedPLIBForCausalLM.postprocess_masks
    for (inp, orig) in [((256, 256), (336, 336)), ((256, 192), (336, 252)), ((256, 40), (300, 47)), ((100, 256), (131, 336)),
                        ((128, 190), (77, 115)), ((64, 64), (64, 64)), ((30, 256), (35, 300))]:
        x = torch.randn(1, 1, 64, 64, generator=g)
        ref = post(None, x, input_size=inp, original_size=orig)
        ora = ops.postprocess_masks(x, inp, orig)
        assert torch.equal(ref, ora), (inp, orig)
        cases.append((x.numpy(), np.array(inp), np.array(orig), ref.numpy()))
    out = {}
    for i, (x, inp, orig, ref) in enumerate(cases):
        out[f"pp{i}_in"] = x; out[f"pp{i}_input_size"] = inp; out[f"pp{i}_original_size"] = orig; out[f"pp{i}_out"] = ref
    out["pp_count"] = np.int64(len(cases))

    # losses on 3 masks of 96x80
    n, H, Wd = 3, 96, 80
    pred = torch.randn(n, 1, H, Wd, generator=g) * 3
    gt = (torch.rand(n, H, Wd, generator=g) > 0.7).float()
    piou = torch.rand(n, 1, generator=g)
    iou_fn, focal_fn = M.MaskIoULoss(), M.FocalLoss()
    ref_terms = []
    for i in range(n):
        gm = gt[i].unsqueeze(0)
        ref_terms.append([M.sigmoid_ce_loss(pred[i], gm, num_masks=1).item(), M.dice_loss(pred[i], gm, num_masks=1).item(),
                          iou_fn(pred[i], gm, piou[i]).item(), focal_fn(pred[i], gm).item()])
        ora = [ops.sigmoid_ce_loss(pred[i], gm, 1).item(), ops.dice_loss(pred[i], gm).item(),
               ops.mask_iou_loss(pred[i], gm, piou[i]).item(), ops.focal_loss(pred[i], gm).item()]
        assert np.allclose(ref_terms[-1], ora, rtol=0, atol=0), (ref_terms[-1], ora)
    out.update(loss_pred=pred.numpy(), loss_gt=gt.numpy(), loss_pred_iou=piou.numpy(), loss_terms=np.array(ref_terms, np.float64))
    # calculate_iou / threshold (train_ds_medplib.py:702-719) restated in ops.threshold_iou; pin the counts
    b, counts, iou, dice = ops.threshold_iou(pred[0, 0], gt[0])
    out.update(thr_mask=b.numpy(), thr_counts=np.array(counts, np.int64), thr_iou=np.float64(iou), thr_dice=np.float64(dice))
    # validate()'s per-sample metrics (train_ds_medplib.py:745-772) from the reference's own intersectionAndUnionGPU
    # (utils/utils.py:92-104): intersection / union per class (K = 2), acc_iou with the "no-object target" rule
    import utils.utils as RU
    vm = []
    for i in range(n):
        output_i = (torch.sigmoid(pred[i]) > 0.1).int()                       # [1, H, W]
        mask_i = gt[i].int().unsqueeze(0)
        # (CPU histc has no int kernel: same values as float)
        inter_i, union_i, _ = RU.intersectionAndUnionGPU(output_i.float().contiguous().clone(), mask_i.float().contiguous(), 2, ignore_index=255)
        acc = inter_i / (union_i + 1e-5)
        acc[union_i == 0] += 1.0
        b = output_i.bool(); gm = mask_i.bool()
        u = torch.logical_or(b, gm).sum()
        iou = 0.0 if u == 0 else (torch.logical_and(b, gm).sum().float() / u.float()).item()
        vm.append(np.concatenate([inter_i.numpy(), union_i.numpy(), acc.numpy(), [iou, 2 * iou / (1 + iou)]]))
        o_b, o_counts, o_iou, o_dice = ops.threshold_iou(pred[i, 0], gt[i])
        om = ops.validate_metrics(o_counts, H * Wd)
        assert np.allclose(np.concatenate([om["intersection"], om["union"], om["acc_iou"], [om["iou"], om["dice"]]]), vm[-1], rtol=1e-6, atol=0), (om, vm[-1])
    out["validate_metrics"] = np.stack(vm)
    np.savez_compressed(os.path.join(OUT, "mask_head_reference.npz"), **out)
    print("mask head goldens ok")


def golden_preprocess():
    """Image preprocessing.  Executed here: the reference's `ResizeLongestSide.get_preprocess_shape`
    (transforms.py:98-108), `LazySupervisedDataset.preprocess` + `pad_tensor_channelwise` (LazySupervisedDataset.py:446-505,
    called unbound on a holder carrying the class constants), the real PIL resize that `ResizeLongestSide.apply_image` ends in
    (torchvision is absent: its one-line glue `resize(to_pil_image(img), (h, w))` is written out as
    `Image.fromarray(img).resize((w, h), BILINEAR)`), and the installed CLIPImageProcessor (transformers 5.15, PIL backend) for
    the rescale + normalise leg."""
    import importlib.util
    import types
    from PIL import Image
    _import_reference_medplib()
    spec = importlib.util.spec_from_file_location("ref_lazy_dataset", os.path.join(REF, "datasets", "LazySupervisedDataset.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    DS = mod.LazySupervisedDataset
    RLS = mod.ResizeLongestSide
    holder = types.SimpleNamespace(pixel_mean=DS.pixel_mean, pixel_std=DS.pixel_std, clip_pixel_mean=DS.clip_pixel_mean,
                                   clip_pixel_std=DS.clip_pixel_std)
    holder.pad_tensor_channelwise = lambda *a, **k: DS.pad_tensor_channelwise(holder, *a, **k)
    from transformers import CLIPImageProcessor
    proc = CLIPImageProcessor(size={"shortest_edge": 336}, crop_size={"height": 336, "width": 336})
    rng = np.random.default_rng(7)
    out = {}
    shapes = [(97, 143), (160, 120), (64, 64), (300, 451)]
    out["n_cases"] = np.int64(len(shapes))
    for i, (h, w) in enumerate(shapes):
        img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        # smooth half of the cases so the resampler sees gradients as well as noise
        if i % 2 == 1:
            yy, xx = np.mgrid[0:h, 0:w]
            img = np.stack([(yy * 255 // max(h - 1, 1)), (xx * 255 // max(w - 1, 1)), ((yy + xx) % 256)], -1).astype(np.uint8)
        mask = (rng.random((h, w)) > 0.6).astype(np.uint8)
        out[f"img{i}"] = img
        out[f"mask{i}"] = mask
        for tag, size in (("sam", 256), ("clip", 336)):
            nh, nw = RLS.get_preprocess_shape(h, w, size)
            resized = np.array(Image.fromarray(img).resize((nw, nh), Image.BILINEAR))
            out[f"{tag}_resized{i}"] = resized
            x = torch.from_numpy(resized).permute(2, 0, 1).contiguous()
            if tag == "sam":
                out[f"sam_out{i}"] = DS.preprocess(holder, x, size).numpy()
            else:
                padded = DS.preprocess(holder, x, size, normalize=False)
                assert padded.dtype == torch.uint8
                out[f"clip_padded{i}"] = padded.numpy()
                out[f"clip_out{i}"] = proc.preprocess(padded, return_tensors="pt")["pixel_values"][0].numpy()
        nh, nw = RLS.get_preprocess_shape(h, w, 336)
        rm = np.array(Image.fromarray(mask).resize((nw, nh), Image.BILINEAR))
        out[f"region_mask{i}"] = DS.preprocess(holder, torch.from_numpy(rm).contiguous(), 336, normalize=False, is_mask=True).numpy()
    # the restatement against what was just executed
    from . import preprocess as P
    for i, (h, w) in enumerate(shapes):
        img, mask = out[f"img{i}"], out[f"mask{i}"]
        s, rs = P.preprocess_sam(img)
        assert np.array_equal(s, out[f"sam_out{i}"]) and tuple(rs) == out[f"sam_resized{i}"].shape[:2]
        assert np.array_equal(P.preprocess_clip(img), out[f"clip_out{i}"])
        assert np.array_equal(P.preprocess_region_mask(mask), out[f"region_mask{i}"])
    np.savez_compressed(os.path.join(OUT, "preprocess_reference.npz"), **out)
    print("preprocess goldens ok")


def dataset_text_cases():
    """Records for the sample-assembly goldens (conversation JSON as the shipped data files hold it)."""
    return [
        {"name": "vqa_single", "has_image": True, "im_start_end": False, "conversations": [
            {"from": "human", "value": "<image>\nWhat modality is this image?"}, {"from": "gpt", "value": "It is a CT scan."}]},
        {"name": "image_tag_last_two_rounds", "has_image": True, "im_start_end": False, "conversations": [
            {"from": "human", "value": "Describe the finding. <image>"}, {"from": "gpt", "value": "A round opacity."},
            {"from": "human", "value": "Segment it."}, {"from": "gpt", "value": "Sure, it is <SEG>."}]},
        {"name": "im_start_end", "has_image": True, "im_start_end": True, "conversations": [
            {"from": "human", "value": "<image>\nIs there a mass?"}, {"from": "gpt", "value": "Yes <SEG>"}]},
        {"name": "region_prompt", "has_image": True, "im_start_end": False, "conversations": [
            {"from": "human", "value": "<image>\nWhat is in <region></region> of the scan?"}, {"from": "gpt", "value": "The liver."}]},
        {"name": "leading_gpt_turn_is_skipped", "has_image": True, "im_start_end": False, "conversations": [
            {"from": "gpt", "value": "ignored"}, {"from": "human", "value": "<image>\nOrgan?"}, {"from": "gpt", "value": "Kidney."}]},
        {"name": "text_only_two_rounds", "has_image": False, "im_start_end": False, "conversations": [
            {"from": "human", "value": "Define pneumothorax."}, {"from": "gpt", "value": "Air in the pleural space."},
            {"from": "human", "value": "Treatment?"}, {"from": "gpt", "value": "Chest drain\nif large."}]},
        {"name": "several_image_tags_collapse", "has_image": True, "im_start_end": False, "conversations": [
            {"from": "human", "value": "Example 1: <image>\nQuery: <image>\nSegment."}, {"from": "gpt", "value": "<SEG>"}]},
        {"name": "answer_with_separator_text_breaks_round", "has_image": True, "im_start_end": False, "conversations": [
            {"from": "human", "value": "<image>\nSay the word."}, {"from": "gpt", "value": "The word is ASSISTANT: ok"}]},
    ]


def golden_dataset():
    """Sample assembly: the reference's own preprocess_multimodal / preprocess_v1 / tokenizer_image_token /
    extract_masks_fun / generate_mask_with_sub_component (datasets/LazySupervisedDataset.py) and the ICL record helpers
    (datasets/ICLLazySupervisedDataset.py, called unbound on a holder) executed on the records above with tests/toy_tokenizer.py.
    cv2 is absent: the ONE cv2 call on these paths, `cv2.connectedComponents` (default 8-connectivity), is served by
    scipy.ndimage.label with a 3x3 structure -- the sub-region goldens are therefore "reference control flow + scipy labels"."""
    import importlib
    import json
    import random
    import tempfile
    import types
    from PIL import Image
    from scipy import ndimage
    _import_reference_medplib()
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
    from toy_tokenizer import ToyTokenizer
    pkg = types.ModuleType("refds")
    pkg.__path__ = [os.path.join(REF, "datasets")]
    sys.modules["refds"] = pkg
    L = importlib.import_module("refds.LazySupervisedDataset")
    I = importlib.import_module("refds.ICLLazySupervisedDataset")
    L.cv2.connectedComponents = lambda m: ndimage.label(m, structure=np.ones((3, 3), dtype=np.uint8))[::-1]
    L.conversation_lib.default_conversation = L.conversation_lib.conv_templates["v1"]
    tok = ToyTokenizer()
    # a real sentencepiece model with the Llama settings, trained here on the case texts (deterministic; committed as a data fixture)
    from toy_tokenizer import SentencePieceLlamaLike
    sp_path = os.path.join(OUT, "tiny_llama_like_sp.model")
    if not os.path.exists(sp_path):
        import sentencepiece as spm
        with tempfile.TemporaryDirectory() as td:
            lines = [L.conversation_lib.conv_templates["v1"].system, "USER: ASSISTANT:"]
            for case in dataset_text_cases():
                lines += [t["value"].replace("<image>", " ").replace("<SEG>", " ") for t in case["conversations"]]
            open(os.path.join(td, "c.txt"), "w").write("\n".join(lines * 8))
            spm.SentencePieceTrainer.train(input=os.path.join(td, "c.txt"), model_prefix=os.path.join(td, "m"), vocab_size=420, model_type="bpe",
                                           byte_fallback=True, character_coverage=1.0, unk_id=0, bos_id=1, eos_id=2, pad_id=-1,
                                           add_dummy_prefix=True, normalization_rule_name="identity", remove_extra_whitespaces=False,
                                           split_digits=True, minloglevel=2)
            os.replace(os.path.join(td, "m.model"), sp_path)
    sp_tok = SentencePieceLlamaLike(sp_path)
    out = {"text": [], "text_sp": [], "tags": [], "subregion": [], "icl": [], "overlay": None}
    for key, tk in (("text", tok), ("text_sp", sp_tok)):
        for case in dataset_text_cases():
            args = types.SimpleNamespace(is_multimodal=True, mm_use_im_start_end=case["im_start_end"])
            convs = [json.loads(json.dumps(case["conversations"]))]
            if case["has_image"]:
                convs = L.preprocess_multimodal(convs, args)
            ex = L.preprocess_v1(convs, tk, has_image=case["has_image"])
            out[key].append({"name": case["name"], "placed": convs, "input_ids": ex["input_ids"].tolist(), "labels": ex["labels"].tolist(),
                             "conversations": ex["conversations"], "question": ex["question"], "gt": ex["gt"]})
    # <mask> / <region> tags (files are opened: tiny PNGs in a temp folder)
    with tempfile.TemporaryDirectory() as root:
        os.makedirs(os.path.join(root, "m"))
        Image.fromarray(np.array([[0, 7, 0], [255, 0, 1]], dtype=np.uint8)).save(os.path.join(root, "m", "a_mask.png"))
        Image.fromarray(np.array([[0, 0], [3, 0]], dtype=np.uint8)).save(os.path.join(root, "r.png"))
        src = {"conversations": [{"from": "human", "value": "<image>\nWhat is <region>r.png</region>? Segment it."},
                                 {"from": "gpt", "value": "A cyst <SEG><mask>m/a_mask.png</mask>."}]}
        rec = json.loads(json.dumps(src))
        masks, _ = L.extract_masks_fun(rec, root, pattern=r"<mask>(.*?)</mask>")
        regions, _ = L.extract_masks_fun(rec, root, pattern=r"<region>(.*?)</region>")
        out["tags"].append({"source": src, "after": rec, "masks": [m.tolist() for m in masks], "regions": [m.tolist() for m in regions]})
    # random sub-region of the largest component (24 x 24 patch-grid masks, the stage-IV parameters)
    rng = np.random.default_rng(3)
    grids = []
    g = np.zeros((24, 24), dtype=np.float32); g[3:12, 4:15] = 1; g[18:21, 18:23] = 1; grids.append(g)          # two components
    g = np.zeros((24, 24), dtype=np.float32); g[10:13, 10:12] = 1; grids.append(g)                              # below min_thresh
    grids.append((rng.random((24, 24)) > 0.55).astype(np.float32))                                               # ragged
    grids.append(np.zeros((24, 24), dtype=np.float32))                                                           # empty -> invalid
    for seed, sel in ((0, [0]), (1, [1]), (2, [2]), (3, [0, 2]), (4, [3]), (5, [2, 3]), (6, [3, 0])):
        random.seed(seed)
        subs, ok = L.generate_mask_with_sub_component([grids[k] for k in sel], min_area=0.2, max_area=1, min_thresh=10)
        out["subregion"].append({"seed": seed, "masks": [grids[k].astype(int).tolist() for k in sel],
                                 "subs": [np.asarray(x).astype(int).tolist() for x in subs], "valid": bool(ok),
                                 "next_draw": random.random()})
    # ICL record helpers
    ICL = I.ICLLazySupervisedDataset
    records = [
        {"image1": "e1.png", "mask1": "e1_m.png", "image2": "e2.png", "mask2": "e2_m.png", "image3": "q.png", "mask3": "q_m.png"},
        {"image": "q.png", "target_mask": "q_m.png", "icl_examples": [{"image": "e1.png", "mask": "e1_m.png"}]},
        {"image": "q.png", "mask": "q_m.png", "examples": [{"image": f"e{k}.png", "mask": f"e{k}_m.png"} for k in range(5)],
         "conversations": [{"from": "human", "value": "A: <image>\nB: <image>\nC: <image>\nQ: <image>\nSegment."}, {"from": "gpt", "value": "<SEG>"}]},
        {"image": "q.png", "examples": [{"image": "e0.png", "mask": "e0_m.png"}],
         "conversations": [{"from": "human", "value": "<image>\nSegment."}, {"from": "gpt", "value": "<SEG>"}]},
    ]
    for rec in records:
        for mode in ("overlay", "separate"):
            holder = types.SimpleNamespace(data_args=types.SimpleNamespace(icl_mask_mode=mode))
            for name in ("_mask_mode", "_expected_image_tokens", "_count_image_tokens", "_has_target_mask_tag", "_build_default_conversation"):
                setattr(holder, name, types.MethodType(getattr(ICL, name), holder))
            raw = json.loads(json.dumps(rec))
            ex = ICL._get_flat_icl_examples(holder, raw)
            prepared = ICL._prepare_source(holder, raw, len(ex))
            out["icl"].append({"record": rec, "mode": mode, "examples": ex, "raw_after": raw, "prepared": prepared})
    img = rng.integers(0, 256, (5, 6, 3), dtype=np.uint8)
    m = (rng.random((5, 6)) > 0.5).astype(np.uint8)
    out["overlay"] = {"image": img.tolist(), "mask": m.tolist(), "out": ICL._overlay_mask(None, img, m).tolist()}

    # the restatement (medplib_amd/dataset.py is host logic of the product; checked here against what was just executed)
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from medplib_amd import dataset as D
    for key, tk in (("text", tok), ("text_sp", sp_tok)):
        for case, exp in zip(dataset_text_cases(), out[key]):
            convs = [json.loads(json.dumps(case["conversations"]))]
            if case["has_image"]:
                D.place_image_token(convs, case["im_start_end"])
            assert convs == exp["placed"], case["name"]
            ex = D.build_v1_example(convs, tk, has_image=case["has_image"])
            assert ex["input_ids"].tolist() == exp["input_ids"] and ex["labels"].tolist() == exp["labels"], case["name"]
            assert ex["conversations"] == exp["conversations"] and ex["question"] == exp["question"] and ex["gt"] == exp["gt"]
    json.dump(out, open(os.path.join(OUT, "dataset_reference.json"), "w"), indent=0, separators=(",", ":"))
    print("dataset goldens ok:", {k: (len(v) if isinstance(v, list) else 1) for k, v in out.items()})


# ------------------------------------------------------------------------------------------------------------------------------
# LISAForCausalLM.model_forward end to end: the reference's own class, constructed at tiny dims from a
# seeded HF-layout state dict, run in train mode on seeded batches; the 10-loss dict, the last hidden state, the trainable-tail
# gradients and a few decoder / front-end gradients of `loss.backward()` become tests/golden/lisa_forward_reference.npz.

LISA_CLIP_DIR = "/tmp/medplib_golden/clip-tiny-336"
LISA_LOSS_WEIGHTS = dict(ce_loss_weight=1.0, dice_loss_weight=5.0, bce_loss_weight=1.0, iou_loss_weight=0.5, focal_loss_weight=1.0)
# parameters whose FULL gradient is stored (everything else trainable in the tail is stored as (sum, L2 norm) per tensor)
LISA_FULL_GRADS = ["model.text_hidden_fcs.0.0.weight", "model.text_hidden_fcs.0.0.bias", "model.text_hidden_fcs.0.2.weight",
                   "model.text_hidden_fcs.0.2.bias", "model.visual_model.mask_decoder.mask_tokens.weight",
                   "model.visual_model.mask_decoder.iou_token.weight",
                   "model.visual_model.mask_decoder.output_upscaling.0.weight",
                   "model.visual_model.mask_decoder.output_upscaling.3.weight",
                   "model.visual_model.mask_decoder.output_hypernetworks_mlps.0.layers.2.weight",
                   "model.visual_model.mask_decoder.iou_prediction_head.layers.2.weight",
                   "model.visual_model.mask_decoder.transformer.layers.0.self_attn.q_proj.weight",
                   "model.visual_model.mask_decoder.transformer.layers.1.cross_attn_image_to_token.out_proj.weight",
                   "model.visual_model.mask_decoder.transformer.final_attn_token_to_image.v_proj.weight",
                   # the differentiable LLM side (pins the oracle's autograd path used by the LoRA-training tests)
                   "lm_head.weight", "model.embed_tokens.weight", "model.layers.0.self_attn.q_proj.weight",
                   "model.layers.1.mlp.down_proj.weight", "model.layers.0.input_layernorm.weight", "model.mm_projector.2.weight"]


def lisa_tiny_cfg():
    """Tiny decoder / CLIP widths but the true token geometry: LISA.model_forward hard-codes 575 = 576 - 1 image rows in its
    <SEG> mask (model/LISA.py:321-325), so CLIP stays 336 px / patch 14; SAM-Med2D is whatever build_sam_vit_b builds (12 blocks)."""
    from medplib_amd.model.config import MedPLIBConfig
    return MedPLIBConfig.tiny(clip_image_size=336, moe_enable=False, iou_loss_weight=LISA_LOSS_WEIGHTS["iou_loss_weight"])


def lisa_cases(cfg):
    """name -> batch: (i) the standard batch, (ii) ragged right padding, (iii) valid_mask_bool = [[True],[True,True],[]].
    Pixel inputs are rounded to bf16-representable values so a bf16 trunk sees exactly what the fp32 reference saw."""
    from . import model as OM
    cases = {"std": OM.make_batch(cfg, 2, seed=5), "ragged": OM.make_batch(cfg, 3, seed=6, ragged=True),
             "multimask": OM.make_batch_multimask(cfg, seed=7)}
    for b in cases.values():
        b["images"] = b["images"].to(torch.bfloat16).float()
        b["images_clip"] = b["images_clip"].to(torch.bfloat16).float()
    return cases


def _build_reference_lisa(cfg, W):
    """Appendix C steps 2-4: a random-init CLIP directory written by a stub-free child process, the stand-in modules, then the
    reference's LISAForCausalLM(config, **kwargs) with the seeded weights loaded by name."""
    import subprocess
    if not os.path.exists(os.path.join(LISA_CLIP_DIR, "config.json")):
        os.makedirs(LISA_CLIP_DIR, exist_ok=True)
        code = (
            "import json\nfrom transformers import CLIPVisionConfig, CLIPVisionModel\n"
            f"c = CLIPVisionConfig(image_size={cfg.clip_image_size}, patch_size={cfg.clip_patch_size}, hidden_size={cfg.clip_hidden_size}, "
            f"intermediate_size={cfg.clip_intermediate_size}, num_hidden_layers={cfg.clip_num_layers}, "
            f"num_attention_heads={cfg.clip_num_heads}, layer_norm_eps={cfg.clip_ln_eps}, hidden_act='quick_gelu')\n"
            f"CLIPVisionModel(c).save_pretrained('{LISA_CLIP_DIR}')\n"
            f"json.dump(dict(crop_size={cfg.clip_image_size}, do_center_crop=True, do_normalize=True, do_resize=True, "
            "image_mean=[0.48145466, 0.4578275, 0.40821073], image_std=[0.26862954, 0.26130258, 0.27577711], resample=3, "
            f"size={cfg.clip_image_size}, image_processor_type='CLIPImageProcessor'), open('{LISA_CLIP_DIR}/preprocessor_config.json', 'w'))\n")
        subprocess.run([sys.executable, "-c", code], check=True)
    _import_reference_medplib()
    import model.LISA as RL
    from model.medplib.model.language_model.medplib_llama import LlavaConfig
    torch.Tensor.cuda = lambda self, *a, **k: self            # `.cuda()` is hard-coded in LISA.py:234,316,323 (Appendix B.7)
    hc = LlavaConfig(vocab_size=cfg.vocab_size, hidden_size=cfg.hidden_size, intermediate_size=cfg.intermediate_size,
                     num_hidden_layers=cfg.num_hidden_layers, num_attention_heads=cfg.num_attention_heads,
                     num_key_value_heads=cfg.num_attention_heads, rms_norm_eps=cfg.rms_norm_eps,
                     max_position_embeddings=cfg.max_position_embeddings)
    hc.mm_vision_tower = hc.vision_tower = LISA_CLIP_DIR
    hc.mm_vision_select_layer = cfg.mm_vision_select_layer
    hc.mm_hidden_size = cfg.clip_hidden_size
    hc.mm_projector_type = "mlp2x_gelu"
    hc.max_sample_point = cfg.max_sample_point
    import io
    import contextlib
    with contextlib.redirect_stdout(io.StringIO()):           # the constructors print their whole config
        m = RL.LISAForCausalLM(hc, seg_token_idx=cfg.seg_token_idx, train_mask_decoder=True, out_dim=cfg.out_dim,
                               vision_pretrained=None, use_mm_start_end=True, max_sample_point=cfg.max_sample_point,
                               **LISA_LOSS_WEIGHTS)
    own = m.state_dict()
    sd = {}
    for k, v in W.items():
        # transformers 5.x dropped the `vision_model.` level of CLIPVisionModel's parameter names (4.31 checkpoints have it)
        k2 = k if k in own else k.replace("vision_tower.vision_tower.vision_model.", "vision_tower.vision_tower.")
        assert k2 in own and own[k2].shape == v.shape, (k, tuple(v.shape))
        sd[k2] = v
    missing, unexpected = m.load_state_dict(sd, strict=False)
    assert not unexpected, unexpected
    # what the seeded dict does not carry: prompt-encoder parts the text-only path never touches (point / box / mask embeddings)
    # ... and CLIP's post_layernorm, which hidden_states[-2] never passes through
    assert all(k.startswith("model.visual_model.prompt_encoder.") or ".post_layernorm." in k for k in missing), missing
    return m


def _ref_key(k, own):
    return k if k in own else k.replace("vision_tower.vision_tower.vision_model.", "vision_tower.vision_tower.")


def golden_lisa():
    from . import model as OM
    cfg = lisa_tiny_cfg()
    W = OM.init_hf_weights(cfg, seed=3)
    out = {"weight_seed": np.int64(3), "loss_weights": np.array([LISA_LOSS_WEIGHTS[k] for k in sorted(LISA_LOSS_WEIGHTS)]),
           "weight_checksum": np.float64(sum(float(v.double().sum()) for v in W.values()))}
    tail = [k for k in W if k.startswith("model.text_hidden_fcs.") or k.startswith("model.visual_model.mask_decoder.")]
    # (sum, L2 norm) of the gradient of EVERY tensor loss.backward() reaches: the trainable tail, the whole decoder, lm_head,
    # embed_tokens and the projector (the frozen towers run under no_grad in the reference too)
    stat_keys = tail + [k for k in W if k.startswith(("model.layers.", "model.norm.", "lm_head.", "model.embed_tokens.", "model.mm_projector."))]
    out["tail_keys"] = np.array(tail)
    out["grad_stat_keys"] = np.array(stat_keys)
    for name, b in lisa_cases(cfg).items():
        # A FRESH module per case, called exactly once.  transformers 5.15 collects `output_hidden_states` with forward hooks that the
        # first LlamaModel.forward installs on every child layer — and the CLIP tower is a child of LisaModel, so from the second
        # call on CLIP reports 2 entries per layer and `hidden_states[-2]` (clip_encoder.py:32) silently becomes the LAST layer.
        # transformers 4.31 (the reference's pin) builds the tuple inline; the first call is the one with its semantics.
        m = _build_reference_lisa(cfg, W)
        m.train()                                              # dropout is 0; eval mode breaks the dense class (SURVEY §8c caveat)
        params = dict(m.named_parameters())
        n_hs = len(m.get_model().get_vision_tower().vision_tower(b["images_clip"][:1], output_hidden_states=True).hidden_states)
        assert n_hs == cfg.clip_num_layers + 1, n_hs
        hs = {}
        h = m.model.norm.register_forward_hook(lambda mod, i, o: hs.__setitem__("last", o.detach()))
        ref_masks, orig_pp = [], m.postprocess_masks             # the masks the reference's forward produces (LISA.py:415-421)
        m.postprocess_masks = lambda *a, **k: (lambda r: (ref_masks.append(r.detach()[:, 0]), r)[1])(orig_pp(*a, **k))
        res = m.model_forward(images=b["images"], images_clip=b["images_clip"], input_ids=b["input_ids"], region_masks=[],
                              labels=b["labels"], attention_masks=b["attention_mask"], offset=None, masks_list=b["masks_list"],
                              label_list=b["label_list"], resize_list=b["resize_list"], inference=False, seg_flag=True,
                              valid_mask_bool=b["valid_mask_bool"], valid_region_masks_bool=[])
        h.remove()
        res["loss"].backward()
        out[f"{name}_losses"] = np.array([float(res[k].detach()) for k in ops.LOSS_KEYS], np.float64)
        out[f"{name}_hidden_tail"] = hs["last"][:, -72:].numpy()          # the text tail of the last hidden state (post final norm)
        out[f"{name}_input_checksum"] = np.float64(float(b["images"].double().sum()) + float(b["images_clip"].double().sum())
                                                   + float(b["input_ids"].sum()))
        for k in (LISA_FULL_GRADS if name == "multimask" else LISA_FULL_GRADS[:4]):
            out[f"{name}_grad_{k}"] = params[_ref_key(k, params)].grad.numpy()
        out[f"{name}_grad_stats"] = np.array([[float(params[k].grad.double().sum()), float(params[k].grad.double().norm())]
                                              for k in stat_keys])
        # ---- the restatement next to it
        Wr = {k: (v.clone().requires_grad_() if k in stat_keys else v) for k, v in W.items()}
        ora, inter = OM.model_forward(b, Wr, cfg, training=True, llm_grad=True, return_intermediates=True)
        ora["loss"].backward()
        dl = max(abs(float(ora[k]) - float(res[k])) for k in ops.LOSS_KEYS)
        dh = (inter["hidden"][:, -72:].detach() - hs["last"][:, -72:]).abs().max().item()
        rel = {k: ((Wr[k].grad - params[_ref_key(k, params)].grad).abs().max() / (params[_ref_key(k, params)].grad.abs().max() + 1e-5)).item()   # floor: k_proj.bias gradients are 0 (softmax shift invariance)
               for k in stat_keys}
        dg = max(rel.values())
        if dg >= 2e-2:
            print({k: f"{v:.1e}" for k, v in rel.items() if v >= 1e-3})
        print(f"lisa case {name}: losses", {k: round(float(res[k]), 5) for k in ("loss", "ce_loss", "mask_loss")},
              f"| oracle: max|dloss| {dl:.2e}  max|dhidden| {dh:.2e}  worst relative gradient error {dg:.2e}")
        assert dl < 2e-4 and dh < 2e-3 and dg < 2e-2, name
        assert len(ref_masks) == len(inter["pred_masks"])
        dm = max((r - p.detach()).abs().max().item() for r, p in zip(ref_masks, inter["pred_masks"]))
        print(f"   masks: {len(ref_masks)} of shapes {[tuple(r.shape) for r in ref_masks]}, max|reference - oracle| {dm:.2e}")
        assert dm < 1e-3
        out[f"{name}_pred_masks"] = np.concatenate([r.reshape(-1).numpy() for r in ref_masks]).astype(np.float32)   # (fp16 until round 3: the cut checks want the logits themselves)
    np.savez_compressed(os.path.join(OUT, "lisa_forward_reference.npz"), **out)
    print("lisa goldens ok:", os.path.getsize(os.path.join(OUT, "lisa_forward_reference.npz")) // 1024, "KiB")


EVAL_CASES = ("fallback", "seg_prompt_two", "seg_generated", "eos")


def evaluate_case(cfg, W, name, edit=None):
    """-> (batch, W_case, max_new_tokens, edit) of one `evaluate()` golden case (B = 1, the reference evaluates one sample at a time,
    vqa_infer.py:528).  fallback: 40-token prompt without <SEG>, nothing generated is <SEG> -> the `[-2:-1]` rule (LISA.py:516-517);
    seg_prompt_two: two <SEG> in the prompt -> the first one (LISA.py:514-515); seg_generated / eos: one lm_head row is made a scaled
    copy of the row of a token the greedy decode emits (`edit` = (dst, src, scale), found by make_golden with the oracle and stored
    in the fixture), so the model itself GENERATES <SEG> / stops at EOS."""
    from . import model as OM
    b = OM.make_batch(cfg, 1, seed=16)        # (of seeds 11..30 the one whose greedy decodes have the widest top-2 logit gaps: >= 0.03)
    b["images"] = b["images"].to(torch.bfloat16).float()
    b["images_clip"] = b["images_clip"].to(torch.bfloat16).float()
    ids = b["input_ids"]
    n_new = {"fallback": 20, "seg_prompt_two": 16, "seg_generated": 18, "eos": 20}[name]
    if name == "seg_prompt_two":
        ids = ids.clone(); ids[0, 50] = cfg.seg_token_idx
    else:
        ids = ids[:, :40].clone()
    b = dict(b, input_ids=ids, labels=None, attention_mask=None)
    Wc = W
    if edit is not None:
        dst, src, scale = int(edit[0]), int(edit[1]), float(edit[2])
        Wc = dict(W)
        lm = W["lm_head.weight"].clone()
        lm[dst] = (scale * lm[src]).to(torch.bfloat16).float()
        Wc["lm_head.weight"] = lm
    return b, Wc, n_new, edit


def golden_evaluate():
    """`LISAForCausalLM.evaluate` (model/LISA.py:473-555 — the dense twin of MedPLIB.py:574-680) EXECUTED: HF greedy `generate`
    driven by the reference's own `prepare_inputs_for_generation`, hidden states of the last step, the <SEG> pick rules, prompt encoder,
    mask decoder, postprocess.  transformers 5.15 here vs the reference's 4.31 pin: `use_cache` is switched off on the instance's
    generation config, which is what the dense class's forward amounts to under 4.31 (it returns `past_key_values=None`,
    medplib_llama.py:143, so every step re-runs the grown sequence and `outputs.hidden_states[-1]` covers all positions)."""
    import contextlib
    import io
    from . import model as OM
    cfg = lisa_tiny_cfg()
    W = OM.init_hf_weights(cfg, seed=3)
    out = {"weight_seed": np.int64(3), "cases": np.array(EVAL_CASES)}
    base_b, _, _, _ = evaluate_case(cfg, W, "fallback")
    base_ids, _ = OM.evaluate(base_b, W, cfg, max_new_tokens=20)
    gen = base_ids[0, 40:].tolist()
    edits = {"seg_generated": (cfg.seg_token_idx, gen[5], 1.25), "eos": (2, gen[8], 1.25)}
    for name in EVAL_CASES:
        b, Wc, n_new, edit = evaluate_case(cfg, W, name, edits.get(name))
        m = _build_reference_lisa(cfg, Wc)           # a fresh module per case (golden_lisa explains why)
        m.eval()
        m.generation_config.use_cache = False
        m.config.use_cache = False
        with contextlib.redirect_stdout(io.StringIO()):           # evaluate() prints the embedding shape
            ref_ids, ref_masks = m.evaluate(b["images_clip"], b["images"], b["input_ids"], b["resize_list"], b["label_list"],
                                            max_new_tokens=n_new)
        ora_ids, ora_masks, dbg = OM.evaluate(b, Wc, cfg, max_new_tokens=n_new, return_debug=True)
        n_in = b["input_ids"].shape[1]
        new = ref_ids[0, n_in:].tolist()
        dm = (ora_masks[0] - ref_masks[0]).abs().max().item()
        print(f"evaluate case {name}: {len(new)} new tokens {new} | oracle ids equal: {torch.equal(ora_ids, ref_ids)}, "
              f"min top-2 gap {min(dbg['gaps']):.3f}, max|mask reference - oracle| {dm:.2e}")
        assert torch.equal(ora_ids, ref_ids) and dm < 1e-4, name
        if name == "seg_generated":
            assert cfg.seg_token_idx in new
        if name == "eos":
            assert new[-1] == 2 and len(new) < n_new
        if name == "fallback":
            assert cfg.seg_token_idx not in ref_ids[0].tolist() and len(new) == n_new
        out[f"{name}_output_ids"] = ref_ids.numpy().astype(np.int64)
        out[f"{name}_pred_mask"] = ref_masks[0].numpy().astype(np.float32)
        out[f"{name}_oracle_top2_gaps"] = np.array(dbg["gaps"], np.float32)
        out[f"{name}_edit"] = np.array(edit if edit is not None else (-1, -1, 0.0), np.float64)
        out[f"{name}_input_checksum"] = np.float64(float(b["images"].double().sum()) + float(b["images_clip"].double().sum())
                                                   + float(b["input_ids"].sum()))
    np.savez_compressed(os.path.join(OUT, "lisa_evaluate_reference.npz"), **out)
    print("evaluate goldens ok:", os.path.getsize(os.path.join(OUT, "lisa_evaluate_reference.npz")) // 1024, "KiB")


def golden_llama_layer():
    """One dense decoder layer + final norm at the 7B dims (d 4096, ff 11008, 32 heads x 128, S 639, one right-padded sample) run by
    the installed HuggingFace `LlamaModel` — the class the reference's dense path instantiates (medplib_llama.py:32-37,98-107) — on
    seeded weights; 64 output rows are stored.  transformers 5.15 here vs the reference's 4.31 pin: same arithmetic."""
    from transformers import LlamaConfig, LlamaModel
    from medplib_amd.model.config import MedPLIBConfig
    from . import llm, model as OM
    cfg = MedPLIBConfig.medplib_7b(num_hidden_layers=1, vocab_size=1024, moe_enable=False, moe_gate_sampling=False)
    W, g = OM.init_decoder_layer_weights(cfg, seed=3)
    emb, kv = OM.decoder_layer_inputs(cfg, g)
    hc = LlamaConfig(vocab_size=cfg.vocab_size, hidden_size=cfg.hidden_size, intermediate_size=cfg.intermediate_size,
                     num_hidden_layers=1, num_attention_heads=cfg.num_attention_heads, num_key_value_heads=cfg.num_attention_heads,
                     rms_norm_eps=cfg.rms_norm_eps, rope_theta=cfg.rope_theta, max_position_embeddings=cfg.max_position_embeddings,
                     attention_bias=False, hidden_act="silu")
    hc._attn_implementation = "eager"
    hf = LlamaModel(hc).eval()
    sd = {k[len("model."):]: v for k, v in W.items() if k.startswith("model.")}
    missing, unexpected = hf.load_state_dict(sd, strict=False)
    assert not unexpected and all("rotary" in m_ or "inv_freq" in m_ for m_ in missing), (missing, unexpected)
    B, S = emb.shape[:2]
    with torch.no_grad():
        ref = hf(inputs_embeds=emb.float(), attention_mask=kv.long(), position_ids=torch.arange(S)[None].expand(B, -1)).last_hidden_state
        ora, _ = llm.llama_forward(emb.float(), kv, W, cfg, training=True)
    rows = torch.nonzero(kv.view(-1)).flatten()
    rows = rows[torch.linspace(0, len(rows) - 1, 64).long()]
    d = (ora.view(B * S, -1)[rows] - ref.view(B * S, -1)[rows]).abs().max().item()
    print("true-dims dense layer: max|oracle - HF| on the stored rows", d, "absmax", ref.abs().max().item())
    assert d < 5e-4
    np.savez_compressed(os.path.join(OUT, "llama_layer_truedims.npz"), rows=rows.numpy(), hidden_rows=ref.view(B * S, -1)[rows].numpy(),
                        weight_seed=np.int64(3), transformers_version=np.array(__import__("transformers").__version__))


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    which = sys.argv[1:] or ["sam", "mask_head", "glue"]
    if "sam" in which:
        golden_sam()
    if "mask_head" in which:
        golden_mask_head()
    if "glue" in which:
        golden_glue()
    if "collate" in which:
        check_collate_contract()
    if "preprocess" in which:
        golden_preprocess()
    if "dataset" in which:
        golden_dataset()
    if "lisa" in which:
        golden_lisa()
    if "llama_layer" in which:
        golden_llama_layer()
    if "evaluate" in which:
        golden_evaluate()
