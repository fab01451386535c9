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
    sys.path.insert(0, os.path.join(REF, "model", "segment_anything_med2d"))
    import modeling  # noqa: the reference's own package (torch-only)
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


def golden_mask_head():
    """postprocess_masks / losses / metrics: run the reference functions from model/MedPLIB.py.  That module imports
    deepspeed/transformers at import time, so the four loss callables and postprocess_masks are exercised through a
    minimal stub environment (types.ModuleType stand-ins for deepspeed & friends)."""
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
    sys.path.insert(0, REF)
    try:
        import model.MedPLIB as M
    except Exception as e:  # pragma: no cover
        print("reference model.MedPLIB import failed:", repr(e))
        raise
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
    np.savez_compressed(os.path.join(OUT, "mask_head_reference.npz"), **out)
    print("mask head goldens ok")


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    which = sys.argv[1:] or ["sam", "mask_head"]
    if "sam" in which:
        golden_sam()
    if "mask_head" in which:
        golden_mask_head()
