"""Seeded per-sample dicts in the shape the reference datasets emit (LazySupervisedDataset / ICLLazySupervisedDataset __getitem__),
shared by oracle/make_golden.py (which runs the REFERENCE collator on them) and tests/test_host_logic.py."""
import types

import torch


def make_cases():
    g = torch.Generator().manual_seed(3)
    tok = types.SimpleNamespace(pad_token_id=0, model_max_length=20)

    def sample(L, n_masks, n_regions, icl=False, hw=(12, 10)):
        ids = torch.randint(3, 50, (L,), generator=g)
        d = {"tokenizer": tok, "input_ids": ids, "labels": torch.where(torch.arange(L) < L // 2, torch.full((L,), -100), ids),
             "masks": [(torch.rand(*hw, generator=g) > 0.5).float() for _ in range(n_masks)],
             "label": [torch.full(hw, 255.0) for _ in range(n_masks)], "resize": [(8, 6)] * n_masks,
             "region_masks": [torch.rand(n_regions, *hw, generator=g) > 0.7] if n_regions else [],
             "image_sam": torch.randn(3, 8, 8, generator=g), "image_clip": torch.randn(*((2, 3, 6, 6) if icl else (3, 6, 6)), generator=g),
             "conversations": [f"conv{L}a", f"conv{L}b"][: 1 + (L % 2)], "image_path": f"/img/{L}.png", "question": f"q{L}", "gt": None,
             "sampled_classes": ["liver"], "answer_type": "seg"}
        if icl:
            d.update(icl_image_paths=[f"/icl/{L}.png"], icl_image_count=2, mask_images=(torch.rand(1, 1, 8, 8, generator=g) > 0.5).float(),
                     image_token_types=["image", "mask", "image"], image_token_lengths=[4, 2, 4])
        return d
    return {
        "seg_ragged_truncated": [sample(9, 1, 0), sample(26, 2, 0), sample(14, 0, 0)],      # 26 > model_max_length: truncation
        "vqa_only_with_regions": [sample(7, 0, 2), sample(11, 0, 0)],
        "icl": [sample(10, 1, 0, icl=True), sample(13, 1, 0, icl=True)],
    }
