"""Sample assembly end to end on the device (medplib_amd/dataset.py pixel layer): PNG + JSON records -> per-sample dicts ->
collate -> model_forward.  Pixel tensors are checked bit for bit against the numpy oracle of the reference's preprocessing
(oracle/preprocess.py, itself pinned by tests/golden/preprocess_reference.npz) on the same decoded arrays."""
import json
import os
import random
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from toy_tokenizer import ToyTokenizer  # noqa: E402

from medplib_amd import dataset as D  # noqa: E402
from medplib_amd.collate import collate  # noqa: E402
from oracle import preprocess as OP  # noqa: E402

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _write_files(root, rng, n=4, h=97, w=143):
    from PIL import Image
    for k in range(n):
        yy, xx = np.mgrid[0:h, 0:w]
        img = np.stack([(yy * 2 + 31 * k) % 256, (xx * 3) % 256, rng.integers(0, 256, (h, w))], -1).astype(np.uint8)
        Image.fromarray(img).save(os.path.join(root, f"img{k}.png"))
        m = np.zeros((h, w), dtype=np.uint8); m[10 + 5 * k:60, 20:90 + 10 * k] = 255
        Image.fromarray(m).save(os.path.join(root, f"img{k}_mask.png"))
    Image.fromarray(np.zeros((h, w), dtype=np.uint8)).save(os.path.join(root, "empty_mask.png"))


def _records():
    return [
        {"image": "img0.png", "answer_type": "open", "conversations": [
            {"from": "human", "value": "<image>\nSegment the lesion."}, {"from": "gpt", "value": "It is <SEG><mask>img0_mask.png</mask>."}]},
        {"image": "img1.png", "conversations": [
            {"from": "human", "value": "<image>\nWhat is in <region>img1_mask.png</region>?"}, {"from": "gpt", "value": "The liver."}]},
        {"image": "img2.png", "conversations": [
            {"from": "human", "value": "<image>\nWhat is in <region>empty_mask.png</region>?"}, {"from": "gpt", "value": "Nothing."}]},
    ]


def test_supervised_samples_match_the_reference_pipeline(dev, tmp_path):
    from PIL import Image
    rng = np.random.default_rng(11)
    _write_files(str(tmp_path), rng)
    tok = ToyTokenizer()
    ds = D.SupervisedDataset(_records(), tok, str(tmp_path), device=dev)
    assert len(ds) == 3 and ds.lengths[0] == 128 + 7 and ds.modality_lengths[1] > 0
    # -- segmentation sample
    s = ds[0]
    rgb = np.array(Image.open(tmp_path / "img0.png").convert("RGB"))
    sam, rs = OP.preprocess_sam(rgb)
    assert torch.equal(s["image_sam"].cpu(), torch.from_numpy(sam)) and tuple(s["resize"][0]) == tuple(rs)
    assert torch.equal(s["image_clip"].cpu(), torch.from_numpy(OP.preprocess_clip(rgb)))
    m = np.array(Image.open(tmp_path / "img0_mask.png").convert("L")); m[m >= 1] = 1
    assert torch.equal(s["masks"][0], torch.tensor(m, dtype=torch.float)) and s["label"][0].shape == m.shape and float(s["label"][0][0, 0]) == 255
    assert s["answer_type"] == "open" and s["region_masks"] == [] and s["conversations"][0].endswith("ASSISTANT: It is <SEG>.</s>")
    assert int((s["input_ids"] == D.IMAGE_TOKEN_INDEX).sum()) == 1 and int((s["labels"] != D.IGNORE_INDEX).sum()) > 0
    # -- region sample: the sub-region is drawn from the 24 x 24 subsample of the CLIP-geometry mask
    random.seed(5)
    s = ds[1]
    m1 = np.array(Image.open(tmp_path / "img1_mask.png").convert("L")); m1[m1 >= 1] = 1
    grid = OP.preprocess_region_mask(m1)[::14, ::14]
    random.seed(5)
    subs, ok = D.region_subcomponents([grid], min_area=0.2, max_area=1, min_thresh=10)
    assert ok and tuple(s["region_masks"][0].shape) == (1, 24, 24) and np.array_equal(s["region_masks"][0][0].numpy(), np.asarray(subs[0], dtype=np.float32))
    assert int((s["input_ids"] == D.REGION_TOKEN_INDEX).sum()) == 1 and "<region></region>" in s["conversations"][0]
    # -- empty region mask: no loss, stub region (LazySupervisedDataset.py:603-613)
    s = ds[2]
    assert bool((s["labels"] == D.IGNORE_INDEX).all()) and tuple(s["region_masks"][0].shape) == (1, 336, 336) and float(s["region_masks"][0].sum()) == 1600
    with pytest.raises(ValueError):
        D.SupervisedDataset([{"conversations": [{"from": "human", "value": "hi"}, {"from": "gpt", "value": "yo"}]}], tok, str(tmp_path), device=dev)[0]


@pytest.mark.parametrize("mode,encoder", [("overlay", False), ("separate", False), ("separate", True)])
def test_icl_samples(dev, tmp_path, mode, encoder):
    from PIL import Image
    rng = np.random.default_rng(12)
    _write_files(str(tmp_path), rng)
    Image.fromarray((rng.random((40, 50)) > 0.5).astype(np.uint8) * 9).save(tmp_path / "small_mask.png")     # resized to the example's shape
    rec = {"image1": "img0.png", "mask1": "img0_mask.png", "image2": "img1.png", "mask2": "small_mask.png", "image3": "img2.png", "mask3": "img2_mask.png"}
    ds = D.ICLSupervisedDataset([rec], ToyTokenizer(), str(tmp_path), device=dev, mask_mode=mode, mask_encoder=encoder)
    s = ds[0]
    n_img = {"overlay": 3, "separate": 5}[mode] - (2 if encoder else 0)
    assert tuple(s["image_clip"].shape) == (n_img, 3, 336, 336) and s["icl_image_count"] == n_img
    kinds = {"overlay": ["image"] * 3, "separate": (["image", "mask"] if encoder else ["image", "image"]) * 2 + ["image"]}[mode]
    assert s["image_token_types"] == kinds and s["image_token_lengths"] == [64 if k == "mask" else 576 for k in kinds]
    assert s["image_path"].endswith("img2.png") and len(s["masks"]) == 1 and s["conversations"][0].endswith("ASSISTANT: <SEG></s>")
    rgb0 = np.array(Image.open(tmp_path / "img0.png").convert("RGB"))
    m0 = (np.array(Image.open(tmp_path / "img0_mask.png").convert("L")) >= 1).astype(np.uint8)
    first = D.overlay_mask(rgb0, m0) if mode == "overlay" else rgb0
    assert torch.equal(s["image_clip"][0].cpu(), torch.from_numpy(OP.preprocess_clip(first)))
    rgb2 = np.array(Image.open(tmp_path / "img2.png").convert("RGB"))
    assert torch.equal(s["image_clip"][-1].cpu(), torch.from_numpy(OP.preprocess_clip(rgb2)))
    if mode == "separate" and not encoder:
        grey = np.stack([m0 * 255] * 3, -1).astype(np.uint8)
        assert torch.equal(s["image_clip"][1].cpu(), torch.from_numpy(OP.preprocess_clip(grey)))
    if encoder:
        assert tuple(s["mask_images"].shape) == (2, 1, 336, 336)
        assert torch.equal(s["mask_images"][0, 0].cpu(), torch.from_numpy((OP.preprocess_region_mask(m0 * 255) > 0).astype(np.float32)))
        small = np.array(Image.open(tmp_path / "small_mask.png").convert("L"))
        ys = np.floor(np.arange(97) * (1.0 / (97 / 40))).astype(int); xs = np.floor(np.arange(143) * (1.0 / (143 / 50))).astype(int)
        ms = (small[ys][:, xs] >= 1).astype(np.uint8)
        assert torch.equal(s["mask_images"][1, 0].cpu(), torch.from_numpy((OP.preprocess_region_mask(ms * 255) > 0).astype(np.float32)))
    else:
        assert s["mask_images"].numel() == 0


def test_records_to_training_step(dev, tmp_path):
    """JSON records -> dataset -> CollatedBatches -> engine step on the tiny model (CLIP 56 px): segmentation + region samples."""
    from medplib_amd import engine
    from medplib_amd.model.config import MedPLIBConfig
    from medplib_amd.model.medplib import MedPLIBForCausalLM
    from medplib_amd.train import dict_to_device
    rng = np.random.default_rng(13)
    _write_files(str(tmp_path), rng)
    cfg = MedPLIBConfig.tiny(moe_enable=True, sam_depth=2)
    tok = ToyTokenizer(model_max_length=512, vocab_size=cfg.vocab_size, seg_token_idx=cfg.seg_token_idx)
    recs = _records()[:2] + [{"image": "img3.png", "conversations": [
        {"from": "human", "value": "<image>\nFind it."}, {"from": "gpt", "value": "<SEG><mask>img3_mask.png</mask>"}]}]
    json.dump(recs, open(tmp_path / "train.json", "w"))
    ds = D.SupervisedDataset(str(tmp_path / "train.json"), tok, str(tmp_path), device=dev, clip_img_size=cfg.clip_image_size)
    batches = D.CollatedBatches(ds, 3, seed=1)
    assert len(batches) == 1
    random.seed(0)
    b = batches[0]
    assert b["images"].shape == (3, 3, 256, 256) and b["images_clip"].shape == (3, 3, 56, 56) and len(b["masks_list"]) == 2
    assert b["rp_flag"] and len(b["region_masks"]) == 1 and tuple(b["region_masks"][0].shape) == (1, 4, 4)
    random.seed(0)
    b2 = batches[0]
    assert torch.equal(b["input_ids"], b2["input_ids"]) and b["image_paths"] == b2["image_paths"]      # seeded order
    torch.manual_seed(0)
    model = MedPLIBForCausalLM(cfg, device=dev).train()
    eng, _, _, _ = engine.initialize(model=model, model_parameters=model.trainable_parameters(),
                                     config={"train_micro_batch_size_per_gpu": 3, "optimizer": {"type": "AdamW", "params": {"lr": 1e-3}}})
    losses = []
    for _ in range(3):
        out = eng(**dict_to_device(b, dev))
        eng.backward(out["loss"]); eng.step()
        losses.append(float(out["loss"].detach()))
        assert float(out["mask_loss"].detach()) > 0 and float(out["ce_loss"].detach()) > 0
    assert all(np.isfinite(losses)) and losses[-1] < losses[0]


def test_icl_records_to_forward(dev, tmp_path):
    """ICL separate mode with the mask encoder and the token compressor: records -> samples -> collate -> model_forward.  The
    prompt keeps its 2n + 1 `<image>` tags (is_multimodal=False: the reference's preprocess_multimodal would fold them into one)."""
    from medplib_amd.model.config import MedPLIBConfig
    from medplib_amd.model.medplib import MedPLIBForCausalLM
    from medplib_amd.train import dict_to_device
    rng = np.random.default_rng(14)
    _write_files(str(tmp_path), rng)
    cfg = MedPLIBConfig.tiny(moe_enable=True, sam_depth=2, mm_token_compress=True, mm_compressed_token_count=8, icl_mask_encoder=True,
                             mask_encoder_token_count=4)
    tok = ToyTokenizer(model_max_length=512, vocab_size=cfg.vocab_size, seg_token_idx=cfg.seg_token_idx)
    recs = [{"image1": "img0.png", "mask1": "img0_mask.png", "image2": "img1.png", "mask2": "img1_mask.png", "image3": "img2.png", "mask3": "img2_mask.png"},
            {"image": "img3.png", "target_mask": "img3_mask.png", "examples": [{"image": "img1.png", "mask": "img1_mask.png"}]}]
    ds = D.ICLSupervisedDataset(recs, tok, str(tmp_path), device=dev, mask_mode="separate", mask_encoder=True, is_multimodal=False,
                                image_token_len=cfg.mm_compressed_token_count, mask_token_len=cfg.mask_encoder_token_count,
                                clip_img_size=cfg.clip_image_size)
    s0, s1 = ds[0], ds[1]
    assert int((s0["input_ids"] == D.IMAGE_TOKEN_INDEX).sum()) == 5 and int((s1["input_ids"] == D.IMAGE_TOKEN_INDEX).sum()) == 3
    assert tuple(s0["image_clip"].shape) == (3, 3, 56, 56) and tuple(s0["mask_images"].shape) == (2, 1, 56, 56)
    b = collate([s0, s1])
    assert b["icl_image_counts"] == [3, 2] and len(b["mask_images"]) == 2 and isinstance(b["images_clip"], list)
    torch.manual_seed(0)
    model = MedPLIBForCausalLM(cfg, device=dev).train()
    out = model(**dict_to_device(b, dev))
    assert np.isfinite(float(out["loss"].detach())) and float(out["mask_loss"].detach()) > 0 and float(out["ce_loss"].detach()) > 0


def test_records_to_inference_loops(dev, tmp_path):
    """vqa_infer.py control flow from JSON records: inference collator, prompt cut after the last "ASSISTANT:", evaluate / generate,
    answers file."""
    from medplib_amd import infer
    from medplib_amd.model.config import MedPLIBConfig
    from medplib_amd.model.medplib import MedPLIBForCausalLM
    rng = np.random.default_rng(15)
    _write_files(str(tmp_path), rng)
    cfg = MedPLIBConfig.tiny(moe_enable=True, sam_depth=2)
    tok = ToyTokenizer(model_max_length=512, vocab_size=cfg.vocab_size, seg_token_idx=cfg.seg_token_idx)
    ds = D.SupervisedDataset(_records()[:2], tok, str(tmp_path), device=dev, clip_img_size=cfg.clip_image_size)
    val = D.CollatedBatches(ds, 1, shuffle=False, collate_fn=lambda s: collate(s, inference=True))
    stub = tok(" ASSISTANT:", add_special_tokens=False).input_ids[-1]
    b0 = val[0]
    cut = infer.prompt_cut(b0["input_ids"], stub)
    assert b0["inference"] and 0 < cut < b0["input_ids"].shape[1] and int(b0["input_ids"][0, cut - 1]) == stub
    torch.manual_seed(0)
    model = MedPLIBForCausalLM(cfg, device=dev).eval()
    random.seed(1)
    seg_only = D.CollatedBatches(D.SupervisedDataset(_records()[:1], tok, str(tmp_path), device=dev, clip_img_size=cfg.clip_image_size),
                                 1, shuffle=False, collate_fn=lambda s: collate(s, inference=True))
    miou, mdice, per = infer.validate_seg(seg_only, model, dev, max_new_tokens=6, colon_id=stub)     # segmentation records only
    assert 0.0 <= miou <= 1.0 and 0.0 <= mdice <= 1.0 and list(per) == ["img0.png"]
    random.seed(1)
    outs = infer.run_vqa(val, model, dev, max_new_tokens=5, colon_id=stub, answers_file=str(tmp_path / "out" / "answers.jsonl"))
    assert len(outs) == 2 and torch.as_tensor(outs[0]).shape[1] >= cut
    lines = [json.loads(x) for x in open(tmp_path / "out" / "answers.jsonl")]
    assert [r["question_id"] for r in lines] == [0, 1] and lines[0]["image_path"].endswith("img0.png")
    assert lines[0]["prompt"] == ["<image>\nSegment the lesion."] and lines[1]["gt"] == ["The liver."] and len(lines[0]["output_ids"]) <= 5


def test_overlay_kernel_equals_the_reference_arithmetic(dev, golden_dir):
    from medplib_amd import preprocess as P
    o = json.load(open(os.path.join(golden_dir, "dataset_reference.json")))["overlay"]        # the reference's _overlay_mask, executed
    img, m = np.array(o["image"], dtype=np.uint8), np.array(o["mask"], dtype=np.uint8)
    got = P.overlay_mask(torch.from_numpy(img).to(dev), torch.from_numpy(m).to(dev))
    assert got.cpu().numpy().tolist() == o["out"]
    rng = np.random.default_rng(2)                                                              # every byte value, ragged size
    img = rng.integers(0, 256, (131, 77, 3), dtype=np.uint8); img.reshape(-1)[:256] = np.arange(256)
    m = (rng.random((131, 77)) > 0.4).astype(np.uint8) * rng.integers(1, 256, (131, 77)).astype(np.uint8)
    got = P.overlay_mask(torch.from_numpy(img).to(dev), torch.from_numpy(m).to(dev))
    assert np.array_equal(got.cpu().numpy(), D.overlay_mask(img, m))


def test_entry_points_on_json_records(dev, tmp_path, monkeypatch):
    """train.py / infer.py with `--dataset medplib_amd.dataset:{from_args,val_from_args}` on JSON files (the tokenizer files are
    not in the image: `load_tokenizer` is replaced by the toy tokenizer, everything else is the shipped code path)."""
    from medplib_amd import infer, train
    from medplib_amd.model.config import MedPLIBConfig
    rng = np.random.default_rng(16)
    _write_files(str(tmp_path), rng)
    cfg = MedPLIBConfig.tiny()
    tok = ToyTokenizer(model_max_length=512, vocab_size=cfg.vocab_size, seg_token_idx=cfg.seg_token_idx)
    monkeypatch.setattr(D, "load_tokenizer", lambda *a, **k: tok)
    recs = [r for r in _records()[:2]] + [{"image": f"img{k}.png", "conversations": [
        {"from": "human", "value": "<image>\nFind it."}, {"from": "gpt", "value": f"<SEG><mask>img{k}_mask.png</mask>"}]} for k in (2, 3)]
    json.dump(recs, open(tmp_path / "train.json", "w"))
    json.dump(recs[2:], open(tmp_path / "val.json", "w"))
    common = ["--model_size", "tiny", "--image_folder", str(tmp_path), "--tokenizer_path", "unused"]
    random.seed(0)
    hist = train.main(common + ["--dataset", "medplib_amd.dataset:from_args", "--data_path", str(tmp_path / "train.json"),
                                "--val_data_path", str(tmp_path / "val.json"), "--log_dir", str(tmp_path / "run"), "--steps_per_epoch", "4",
                                "--batch_size", "2", "--lr", "1e-3"])
    assert len(hist) == 4 and all(np.isfinite(hist))
    assert os.path.exists(tmp_path / "run" / "ckpt_model" / "latest")
    stub = tok(" ASSISTANT:", add_special_tokens=False).input_ids[-1]
    out = infer.main(common + ["--dataset", "medplib_amd.dataset:val_from_args", "--val_data_path", str(tmp_path / "val.json"),
                               "--max_new_tokens", "4", "--colon_token_id", str(stub), "--eval_vqa", "--answers_file", str(tmp_path / "ans.jsonl")])
    assert 0.0 <= out["miou"] <= 1.0 and len(out["vqa_output_ids"]) == 2
    assert len(open(tmp_path / "ans.jsonl").read().splitlines()) == 2
