"""Sample assembly in front of the collator: JSON record -> the per-sample dict `collate()` consumes.

Two layers:

* text (pure host logic, no torch kernels): the v1 conversation template, `<image>` placement, tokenisation with the image /
  region sentinels, target masking, `<mask>` / `<region>` tag extraction, the ICL conversation builder.  Mirrors
  datasets/LazySupervisedDataset.py:89-233, 239-272, 353-387 and datasets/ICLLazySupervisedDataset.py:98-178; pinned by
  tests/golden/dataset_reference.json = the reference's own functions executed on the same records with the same tokenizer.
* pixels: `SupervisedDataset` / `ICLSupervisedDataset.__getitem__` decode with PIL and hand uint8 tensors to `preprocess.py`
  (device kernels, bit-exact with the reference's CPU path; no CPU fallback).  The reference decodes with cv2.imread: identical
  for PNG / BMP; JPEG decoders may differ by a count in a few pixels (libjpeg build), which is outside this repository.

The tokenizer is whatever the caller passes (a Hugging Face Llama tokenizer in production): it is only asked for
`tok(text).input_ids`, `tok(text, add_special_tokens=False).input_ids`, `bos_token_id`, `pad_token_id`, `model_max_length`."""
import copy
import json
import os
import random
import re
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

IGNORE_INDEX = -100
IMAGE_TOKEN_INDEX = -200
REGION_TOKEN_INDEX = -300
IMAGE_TAG = "<image>"
IM_START, IM_END = "<im_start>", "<im_end>"

# the "v1" (Vicuna) template the training / inference entry points select (train_ds_medplib.py:258, conversation.py:275-285)
V1_SYSTEM = ("A chat between a curious user and an artificial intelligence assistant. "
             "The assistant gives helpful, detailed, and polite answers to the user's questions.")
V1_ROLES = ("USER", "ASSISTANT")
V1_SEP, V1_SEP2 = " ", "</s>"
_SPEAKER = {"human": V1_ROLES[0], "gpt": V1_ROLES[1]}


# ------------------------------------------------------------------------------------------------------------------ text layer
def v1_prompt(turns: Sequence[Tuple[str, Optional[str]]]) -> str:
    """Two-separator prompt: system, then `ROLE: text` closed by ' ' after the user and '</s>' after the assistant; an empty
    message leaves the bare `ROLE:` generation stub (Conversation.get_prompt, SeparatorStyle.TWO, conversation.py:53-62)."""
    pieces = [V1_SYSTEM, V1_SEP]
    for k, (role, text) in enumerate(turns):
        pieces.append(f"{role}: {text}{V1_SEP2 if k & 1 else V1_SEP}" if text else f"{role}:")
    return "".join(pieces)


def place_image_token(conversations: List[List[Dict]], mm_use_im_start_end: bool = False) -> List[List[Dict]]:
    """Move every turn's `<image>` tags to one leading `<image>\\n` (optionally wrapped in <im_start>/<im_end>), in place
    (preprocess_multimodal, LazySupervisedDataset.py:109-121).  A turn holding several tags comes out with ONE leading tag --
    that is what the reference's replace('') + prefix does, also to the multi-image ICL prompts it builds (golden case
    `icl_*`); prompts that must keep n tags are assembled with is_multimodal=False."""
    for conv in conversations:
        for turn in conv:
            text = str(turn["value"])
            if IMAGE_TAG not in text:
                continue
            body = (IMAGE_TAG + "\n" + text.replace(IMAGE_TAG, "").strip()).strip()
            tag = IM_START + IMAGE_TAG + IM_END if mm_use_im_start_end else IMAGE_TAG
            turn["value"] = body.replace(IMAGE_TAG, tag)
    return conversations


def tokenize_with_image_tokens(prompt: str, tokenizer, image_token_index: int = IMAGE_TOKEN_INDEX) -> List[int]:
    """Token ids of `prompt` with one IMAGE_TOKEN_INDEX per `<image>` tag and a REGION_TOKEN_INDEX between every adjacent
    `<region>` `</region>` pair (tokenizer_image_token, LazySupervisedDataset.py:353-387)."""
    chunks = [tokenizer(c).input_ids for c in prompt.split(IMAGE_TAG)]
    has_bos = bool(chunks) and len(chunks[0]) > 0 and chunks[0][0] == tokenizer.bos_token_id
    skip = 1 if has_bos else 0
    ids: List[int] = [chunks[0][0]] if has_bos else []
    for k, c in enumerate(chunks):
        if k:
            ids.append(image_token_index)       # the separator is [index] * (skip + 1) with its first `skip` entries dropped
        ids.extend(c[skip:])
    r_open = tokenizer("<region>", add_special_tokens=False).input_ids[0]
    r_close = tokenizer("</region>", add_special_tokens=False).input_ids[0]
    out: List[int] = []
    for k, t in enumerate(ids):
        out.append(t)
        if t == r_open and k + 1 < len(ids) and ids[k + 1] == r_close:
            out.append(REGION_TOKEN_INDEX)
    return out


def build_v1_example(sources: Sequence[Sequence[Dict]], tokenizer, has_image: bool = False) -> Dict:
    """Conversations -> input_ids, labels (only assistant text supervised), prompt strings, questions, answers
    (preprocess_v1, LazySupervisedDataset.py:124-232).  Keeps the reference's length bookkeeping, including the `- 2`
    (BOS + the trailing-space piece of a sentencepiece tokenizer) and the all-IGNORE row on a length mismatch."""
    prompts, questions, answers = [], [], []
    for n, conv in enumerate(sources):
        if _SPEAKER[conv[0]["from"]] != V1_ROLES[0]:
            conv = conv[1:]
        turns = []
        for k, turn in enumerate(conv):
            if turn["from"] == "human":
                questions.append(turn["value"].replace(IM_START + IMAGE_TAG + IM_END + "\n", ""))
            else:
                answers.append(turn["value"])
            assert _SPEAKER[turn["from"]] == V1_ROLES[k % 2], f"conversation {n}: turns must alternate human / gpt"
            turns.append((_SPEAKER[turn["from"]], turn["value"]))
        prompts.append(v1_prompt(turns))

    if has_image:
        input_ids = torch.stack([torch.tensor(tokenize_with_image_tokens(p, tokenizer), dtype=torch.long) for p in prompts], 0)
        count = lambda text: len(tokenize_with_image_tokens(text, tokenizer))
    else:
        input_ids = tokenizer(prompts, return_tensors="pt", padding="longest", max_length=tokenizer.model_max_length,
                              truncation=True).input_ids
        count = lambda text: len(tokenizer(text).input_ids)

    labels = input_ids.clone()
    answer_mark = V1_SEP + V1_ROLES[1] + ": "
    for prompt, row in zip(prompts, labels):
        total = int(row.ne(tokenizer.pad_token_id).sum())
        keep = torch.zeros_like(row, dtype=torch.bool)        # supervised positions
        pos = 1                                               # BOS
        for rnd in prompt.split(V1_SEP2):
            if rnd == "":
                break
            halves = rnd.split(answer_mark)
            if len(halves) != 2:
                break
            n_round = count(rnd)
            n_instr = count(halves[0] + answer_mark) - 2
            keep[pos + n_instr: pos + n_round] = True
            pos += n_round
        keep[pos:] = False
        if pos < tokenizer.model_max_length and pos != total:
            print(f"WARNING: tokenization mismatch: {pos} vs. {total}. (ignored)")
            keep[:] = False
        row[~keep] = IGNORE_INDEX
    return {"input_ids": input_ids, "labels": labels, "conversations": prompts, "question": questions, "gt": answers}


def pull_tagged_files(source: Dict, tag: str) -> List[str]:
    """File names inside `<mask>..</mask>` (tag='mask': the whole tag is removed from the turn, which must hold `<SEG>`) or
    `<region>..</region>` (tag='region': only the name is removed, the empty pair stays for the region sentinel); at most one
    per turn; `source['conversations']` is edited in place (extract_masks_fun, LazySupervisedDataset.py:239-272)."""
    assert tag in ("mask", "region")
    rx = re.compile(f"<{tag}>(.*?)</{tag}>")
    names = []
    for turn in source["conversations"]:
        found = rx.findall(str(turn["value"]))
        if not found:
            continue
        assert len(found) == 1, "Only one mask is supported in one turns."
        names.append(found[0])
        if tag == "mask":
            assert "<SEG>" in turn["value"], "SEG token is required in the answer when exist mask."
            turn["value"] = turn["value"].replace(f"<mask>{found[0]}</mask>", "")
        else:
            turn["value"] = turn["value"].replace(found[0], "")
    return names


def _label8(mask: np.ndarray) -> np.ndarray:
    """8-connected component labels in raster order (the reference calls cv2.connectedComponents, default connectivity 8)."""
    from scipy import ndimage
    return ndimage.label(mask.astype(np.uint8), structure=np.ones((3, 3), dtype=np.uint8))[0]


def _grow_subregion(component: np.ndarray, min_area, max_area, min_thresh, rng) -> np.ndarray:
    """Random depth-first sub-region of one component (generate_sub_connected_component, :274-311): same draws in the same
    order from `rng` (uniform for the area ratio, choice for the seed pixel, one shuffle of the 9 neighbours per visited pixel)."""
    area = int(np.sum(component == 1))
    if area < min_thresh:
        return component
    want = 0
    while want // min_thresh < 1:
        want = int(area * rng.uniform(min_area, max_area))
    grown = np.zeros_like(component)
    ys, xs = np.where(component == 1)
    todo = [rng.choice(list(zip(ys, xs)))]
    n_set = 0
    H, W = component.shape
    while todo:
        y, x = todo.pop()
        if grown[y, x] == 0:
            n_set += 1
        grown[y, x] = 1
        if n_set >= want:
            break
        around = [(y + dy, x + dx) for dy in (-1, 0, 1) for dx in (-1, 0, 1)]
        rng.shuffle(around)
        for (ny, nx) in around:
            if 0 <= ny < H and 0 <= nx < W and component[ny, nx] == 1 and grown[ny, nx] == 0:
                todo.append((ny, nx))
    return grown


def region_subcomponents(masks: Sequence[np.ndarray], min_area=0.4, max_area=1.0, min_thresh=1000, rng=random):
    """Per region mask: a random sub-region of its largest 8-connected component; (sub_masks, is_valid)
    (generate_mask_with_sub_component, :313-349).  Faithful to the reference's control flow: the upper ratio bound is
    overwritten by the running largest AREA, a sub-region is drawn at every label visited (consuming `rng`), the last one is
    kept; an empty mask yields an all-ones 336 x 336 plane and `is_valid = False` (the flag reflects the LAST mask)."""
    out, valid = [], False
    for m in masks:
        m = np.array(m)
        if np.sum(m) > 0:
            labels = _label8(m)
            best_area, best = 0, 0
            for lv in np.unique(labels)[1:]:
                a = int(np.sum(labels == lv))
                if a > best_area:
                    best_area, best = a, lv
                valid = True
                max_area = best_area
                sub = _grow_subregion(np.where(labels == best, 1, 0), min_area, max_area, min_thresh, rng)
        else:
            valid = False
            sub = np.ones((336, 336))
        out.append(sub)
    return out, valid


# ---- ICL records (ICLLazySupervisedDataset.py:98-178)
def icl_examples_of(source: Dict) -> List[Dict[str, str]]:
    """Up to three {image, mask} in-context examples: an explicit `icl_examples` / `examples` list, or the flat
    image1/mask1 .. imageN/maskN layout whose last index is the query when no `image` key exists (the record is completed in
    place with `image` / `target_mask`, as the reference does)."""
    listed = source.get("icl_examples", source.get("examples", []))
    if listed:
        return listed[:3]
    idx = sorted(int(k[5:]) for k in source if k.startswith("image") and k[5:].isdigit())
    if not idx:
        return []
    query = None
    if "image" not in source:
        query = idx[-1]
        source.setdefault("image", source[f"image{query}"])
        if f"mask{query}" in source:
            source.setdefault("target_mask", source[f"mask{query}"])
    return [{"image": source[f"image{k}"], "mask": source[f"mask{k}"]} for k in idx
            if k != query and f"mask{k}" in source][:3]


def _icl_target_mask(source):
    return source.get("target_mask", source.get("mask", source.get("mask3", None)))


def icl_default_conversation(source: Dict, n_examples: int, mask_mode: str) -> List[Dict[str, str]]:
    lines = []
    for k in range(1, n_examples + 1):
        lines.append(f"Example {k} image: <image>\nExample {k} mask: <image>" if mask_mode == "separate" else
                     f"Example {k}: <image>\nThe blue overlay is the reference segmentation mask.")
    lines.append("Query: <image>\nRefer to the previous examples and segment the corresponding target in this image.")
    answer = "<SEG>"
    tm = _icl_target_mask(source)
    if tm is not None:
        answer += f"<mask>{tm}</mask>"
    return [{"from": "human", "value": "\n".join(lines)}, {"from": "gpt", "value": answer}]


def icl_prepare_source(source: Dict, n_examples: int, mask_mode: str) -> Dict:
    """A deep copy whose conversation carries enough `<image>` tags (else the default ICL prompt) and the target `<mask>` tag."""
    assert mask_mode in ("overlay", "separate"), f"Unsupported ICL mask mode: {mask_mode}"
    src = copy.deepcopy(source)
    need = 2 * n_examples + 1 if mask_mode == "separate" else n_examples + 1
    have = sum(str(t.get("value", "")).count(IMAGE_TAG) for t in src.get("conversations", []))
    if "conversations" not in src or have < need:
        src["conversations"] = icl_default_conversation(src, n_examples, mask_mode)
    elif not any(re.search(r"<mask>(.*?)</mask>", str(t.get("value", ""))) for t in src["conversations"]):
        tm = _icl_target_mask(src)
        if tm is not None:
            src["conversations"][-1]["value"] = str(src["conversations"][-1]["value"]) + f"<mask>{tm}</mask>"
    return src


def overlay_mask(image_rgb: np.ndarray, mask: np.ndarray) -> np.ndarray:
    """Blue 55 % overlay of the reference mask on an example image, float32 arithmetic, truncation to uint8 (:46-50).  Host
    statement of `preprocess.overlay_mask` (the device kernel the dataset uses); kept for the goldens and as documentation."""
    tint = np.array([118, 158, 224], dtype=np.float32)
    img = image_rgb.astype(np.float32)
    on = mask > 0
    img[on] = img[on] * 0.45 + tint * 0.55
    return np.clip(img, 0, 255).astype(np.uint8)


# ---------------------------------------------------------------------------------------------------------------- pixel layer
def _open_rgb(path: str) -> np.ndarray:
    from PIL import Image
    return np.array(Image.open(path).convert("RGB"))


def _open_gray(path: str) -> np.ndarray:
    from PIL import Image
    return np.array(Image.open(path).convert("L"))


class SupervisedDataset(torch.utils.data.Dataset):
    """JSON records {image, conversations[, answer_type]} -> collate-ready samples (LazySupervisedDataset.__getitem__, :505-617).
    Pixel work runs on `device` through `preprocess.py`; tensors stay there (the collator stacks them in HBM)."""
    ignore_label = 255
    sam_img_size = 256
    clip_img_size = 336

    def __init__(self, data, tokenizer, image_folder: str, device="cuda", is_multimodal: bool = True,
                 mm_use_im_start_end: bool = False, sam_img_size: int = 256, clip_img_size: int = 336, rng=random):
        self.records = json.load(open(data)) if isinstance(data, str) else list(data)
        self.tokenizer, self.image_folder, self.device = tokenizer, image_folder, torch.device(device)
        self.is_multimodal, self.mm_use_im_start_end, self.rng = is_multimodal, mm_use_im_start_end, rng
        self.sam_img_size, self.clip_img_size = sam_img_size, clip_img_size      # 256 / 336 in every shipped script

    def __len__(self):
        return len(self.records)

    @property
    def lengths(self):
        return [sum(len(t["value"].split()) for t in r["conversations"]) + (128 if "image" in r else 0) for r in self.records]

    @property
    def modality_lengths(self):
        out = []
        for r in self.records:
            n = sum(len(t["value"].split()) for t in r["conversations"])
            out.append(n if "image" in r else -n)
        return out

    def resolve(self, name: str) -> str:
        if os.path.exists(name):
            return name
        if "llavamed" in name:
            return os.path.join("/".join(self.image_folder.split("/")[:-1]), name)
        return os.path.join(self.image_folder, name)

    def _dev(self, arr: np.ndarray) -> torch.Tensor:
        return torch.from_numpy(np.ascontiguousarray(arr)).to(self.device)

    def _load_binary(self, name: str) -> np.ndarray:
        m = _open_gray(os.path.join(self.image_folder, name))
        m[m >= 1] = 1
        return m

    def _region_planes(self, names: Sequence[str]):
        """<region> masks: CLIP-geometry resize + pad (device), 1/14 nearest subsample to the 24 x 24 patch grid
        (cv2.resize fx = fy = 1/14, INTER_NEAREST = every 14th pixel), random sub-region of the largest component (:516-521)."""
        from . import preprocess as P
        grids = [P.preprocess_region_mask(self._dev(self._load_binary(n)), self.clip_img_size)[::14, ::14].cpu().numpy() for n in names]
        return region_subcomponents(grids, min_area=0.2, max_area=1, min_thresh=10, rng=self.rng)

    def __getitem__(self, i) -> Dict:
        from . import preprocess as P
        rec = copy.deepcopy(self.records[i])
        mask_names = pull_tagged_files(rec, "mask")
        region_names = pull_tagged_files(rec, "region")
        masks = [torch.tensor(self._load_binary(n), dtype=torch.float) for n in mask_names]
        regions, regions_ok = self._region_planes(region_names)
        if "image" not in rec:
            raise ValueError("text-only records are not supported: the reference's __getitem__ needs an image for the SAM branch")
        path = self.resolve(rec["image"])
        assert os.path.exists(path), f"{path} dose not exist"
        rgb = self._dev(_open_rgb(path))
        image_sam, resize = P.preprocess_sam(rgb, self.sam_img_size)
        image_clip = P.preprocess_clip(rgb, self.clip_img_size)
        convs = [copy.deepcopy(rec["conversations"])]
        if self.is_multimodal:
            place_image_token(convs, self.mm_use_im_start_end)
        ex = build_v1_example(convs, self.tokenizer, has_image=True)
        out = {"input_ids": ex["input_ids"][0], "labels": ex["labels"][0], "conversations": ex["conversations"],
               "question": ex["question"], "gt": ex["gt"], "image_clip": image_clip, "masks": masks,
               "region_masks": [torch.tensor(np.asarray(r), dtype=torch.float).unsqueeze(0) for r in regions],
               "image_sam": image_sam, "image_path": path, "inference": False, "tokenizer": self.tokenizer,
               "answer_type": rec.get("answer_type", None)}
        if masks:
            out["label"] = [torch.ones(masks[0].shape[0], masks[0].shape[1]) * self.ignore_label] * len(masks)
            out["resize"] = [resize] * len(masks)
        if regions and not regions_ok:          # an empty region mask: the sample is kept but carries no loss (:603-613)
            out["labels"] = torch.full_like(out["labels"], IGNORE_INDEX)
            stub = torch.zeros(1, 336, 336)
            stub[:, :40, :40] = 1
            out["region_masks"] = [stub]
        return out


class ICLSupervisedDataset(SupervisedDataset):
    """MedPLIB-ICL records: 1-3 in-context (image, mask) examples + the query (ICLLazySupervisedDataset.__getitem__, :180-266).
    mask_mode 'overlay': each example is one CLIP image with the mask tinted in; 'separate': image and mask are two `<image>`
    slots, the mask either rendered as a grey RGB CLIP image or (mask_encoder=True) kept as a binary 336 x 336 plane for the
    MaskTokenEncoder."""

    def __init__(self, data, tokenizer, image_folder, device="cuda", mask_mode="overlay", mask_encoder=False,
                 image_token_len=576, mask_token_len=64, **kw):
        super().__init__(data, tokenizer, image_folder, device=device, **kw)
        assert mask_mode in ("overlay", "separate"), f"Unsupported ICL mask mode: {mask_mode}"
        self.mask_mode, self.mask_encoder = mask_mode, bool(mask_encoder) and mask_mode == "separate"
        self.image_token_len, self.mask_token_len = image_token_len, mask_token_len

    def _example_mask(self, name: str, shape) -> np.ndarray:
        path = self.resolve(name)
        assert os.path.exists(path), f"{path} dose not exist"
        m = _open_gray(path)
        if m.shape[:2] != tuple(shape):     # cv2.resize(..., INTER_NEAREST): src index = floor(dst * (1 / (dst_size / src_size)))
            ys = np.minimum(np.floor(np.arange(shape[0]) * (1.0 / (shape[0] / m.shape[0]))).astype(np.int64), m.shape[0] - 1)
            xs = np.minimum(np.floor(np.arange(shape[1]) * (1.0 / (shape[1] / m.shape[1]))).astype(np.int64), m.shape[1] - 1)
            m = m[ys][:, xs]
        return (m >= 1).astype(np.uint8)

    def _encoder_plane(self, mask: np.ndarray) -> torch.Tensor:
        from . import preprocess as P
        plane = P.preprocess_region_mask(self._dev(mask.astype(np.uint8) * 255), self.clip_img_size)
        return (plane > 0).float().unsqueeze(0)

    def __getitem__(self, i) -> Dict:
        from . import preprocess as P
        raw = self.records[i]
        examples = icl_examples_of(raw)
        assert 1 <= len(examples) <= 3, "MedPLIB-ICL requires 1 to 3 in-context examples."
        rec = icl_prepare_source(raw, len(examples), self.mask_mode)
        mask_names = pull_tagged_files(rec, "mask")
        masks = [torch.tensor(self._load_binary(n), dtype=torch.float) for n in mask_names]
        target = rec.get("image", rec.get("image3", None))
        assert target is not None, "MedPLIB-ICL requires a target image in `image` or `image3`."
        path = self.resolve(target)
        assert os.path.exists(path), f"{path} dose not exist"
        target_rgb = _open_rgb(path)
        image_sam, resize = P.preprocess_sam(self._dev(target_rgb), self.sam_img_size)
        clip = lambda rgb: P.preprocess_clip(self._dev(rgb), self.clip_img_size)

        clips, planes, kinds, lens, paths = [], [], [], [], []
        for ex in examples:
            ex_path = self.resolve(ex["image"])
            assert os.path.exists(ex_path), f"{ex_path} dose not exist"
            ex_rgb = _open_rgb(ex_path)
            ex_mask = self._example_mask(ex["mask"], ex_rgb.shape[:2])
            if self.mask_mode == "separate":
                clips.append(clip(ex_rgb)); kinds.append("image"); lens.append(self.image_token_len)
                if self.mask_encoder:
                    planes.append(self._encoder_plane(ex_mask)); kinds.append("mask"); lens.append(self.mask_token_len)
                else:
                    grey = (ex_mask * 255).astype(np.uint8)
                    clips.append(clip(np.stack([grey, grey, grey], -1))); kinds.append("image"); lens.append(self.image_token_len)
                paths += [ex_path, self.resolve(ex["mask"])]
            else:
                tinted = P.overlay_mask(self._dev(ex_rgb), self._dev(ex_mask))              # device twin of overlay_mask()
                clips.append(P.preprocess_clip(tinted, self.clip_img_size)); kinds.append("image"); lens.append(self.image_token_len)
                paths.append(ex_path)
        clips.append(clip(target_rgb)); kinds.append("image"); lens.append(self.image_token_len)

        convs = [copy.deepcopy(rec["conversations"])]
        if self.is_multimodal:
            place_image_token(convs, self.mm_use_im_start_end)
        ex = build_v1_example(convs, self.tokenizer, has_image=True)
        out = {"input_ids": ex["input_ids"][0], "labels": ex["labels"][0], "conversations": ex["conversations"],
               "question": ex["question"], "gt": ex["gt"], "image_clip": torch.stack(clips, 0), "image_sam": image_sam,
               "image_path": path, "icl_image_paths": paths, "icl_image_count": len(clips),
               "mask_images": torch.stack(planes, 0) if planes else torch.empty(0), "image_token_types": kinds,
               "image_token_lengths": lens, "masks": masks, "region_masks": [], "inference": False,
               "tokenizer": self.tokenizer, "answer_type": rec.get("answer_type", None)}
        if masks:
            out["label"] = [torch.ones(masks[0].shape[0], masks[0].shape[1]) * self.ignore_label] * len(masks)
            out["resize"] = [resize] * len(masks)
        return out


# ------------------------------------------------------------------------------------------------------- batches for train.py
class LazySupervisedDataset(SupervisedDataset):
    """The reference's constructor signature (datasets/LazySupervisedDataset.py:400-420, called at train_ds_medplib.py:368):
    `LazySupervisedDataset(data_path, tokenizer, data_args, sam_img_size)` with `data_args` the driver's SimpleNamespace (:354-366)."""

    def __init__(self, data_path, tokenizer, data_args, sam_img_size=256, device="cuda"):
        proc = getattr(data_args, "image_processor", None)
        crop = getattr(proc, "crop_size", None)
        clip = int(crop["height"] if isinstance(crop, dict) else (crop or 336))
        super().__init__(data_path, tokenizer, data_args.image_folder, device=device, is_multimodal=getattr(data_args, "is_multimodal", True),
                         mm_use_im_start_end=getattr(data_args, "mm_use_im_start_end", False), sam_img_size=sam_img_size, clip_img_size=clip)
        self.data_args = data_args


class ICLLazySupervisedDataset(ICLSupervisedDataset):
    """`ICLLazySupervisedDataset(data_path, tokenizer, data_args, sam_img_size)` (datasets/ICLLazySupervisedDataset.py:19-60)."""

    def __init__(self, data_path, tokenizer, data_args, sam_img_size=256, device="cuda"):
        proc = getattr(data_args, "image_processor", None)
        crop = getattr(proc, "crop_size", None)
        clip = int(crop["height"] if isinstance(crop, dict) else (crop or 336))
        n_img = data_args.mm_compressed_token_count if getattr(data_args, "mm_token_compress", False) else (clip // 14) ** 2
        super().__init__(data_path, tokenizer, data_args.image_folder, device=device, mask_mode=getattr(data_args, "icl_mask_mode", "overlay"),
                         mask_encoder=getattr(data_args, "icl_mask_encoder", False), image_token_len=n_img,
                         mask_token_len=getattr(data_args, "mask_encoder_token_count", 64), is_multimodal=False,
                         mm_use_im_start_end=getattr(data_args, "mm_use_im_start_end", False), sam_img_size=sam_img_size, clip_img_size=clip)
        self.data_args = data_args


class CollatedBatches(torch.utils.data.Dataset):
    """Batch-indexed view for `train.py` (`data[it]` = one collated micro-batch).  The samples are walked in a per-epoch
    permutation (seeded: the same on every rank and in every run) in strides of world * B; rank r takes the r-th B of each
    stride, so the ranks' shards are disjoint like a DistributedSampler's (train_ds_medplib.py:430-450); the walk wraps around
    the end of the data into the next epoch's permutation."""

    def __init__(self, samples, batch_size: int, seed: int = 0, shuffle: bool = True, collate_fn=None, rank: int = 0, world: int = 1):
        from .collate import collate
        self.samples, self.B, self.seed, self.shuffle = samples, int(batch_size), int(seed), shuffle
        self.rank, self.world = int(rank), int(world)
        self.collate = collate_fn or collate
        self._perm = (None, None)

    def __len__(self):
        return (len(self.samples) + self.B * self.world - 1) // (self.B * self.world)

    def _order(self, epoch):
        if self._perm[0] != epoch:
            n = len(self.samples)
            order = torch.randperm(n, generator=torch.Generator().manual_seed(self.seed + epoch)).tolist() if self.shuffle else list(range(n))
            self._perm = (epoch, order)
        return self._perm[1]

    def indices(self, i):
        """Sample indices of this rank's i-th micro-batch."""
        n = len(self.samples)
        first = (i * self.world + self.rank) * self.B
        picks = []
        for j in range(first, first + self.B):
            epoch, k = divmod(j, n)
            picks.append(self._order(epoch)[k])
        return picks

    def __getitem__(self, i):
        return self.collate([self.samples[k] for k in self.indices(i)])


def load_tokenizer(path: str, model_max_length: int = 512, use_mm_start_end: bool = False, extra_tokens: Sequence[str] = ()):
    """The reference's tokenizer setup (train_ds_medplib.py:197-216): slow Llama tokenizer, right padding, pad = unk, the added
    special tokens (`utils.ADD_OTHERS_TOKENS` -- `<SEG>`, `<region>`, `</region>`, ... -- passed in `extra_tokens`, then
    `<gen_1>` .. `<gen_256>`), optionally <im_start>/<im_end>.  Needs the tokenizer files on disk (no network here)."""
    import transformers
    tok = transformers.AutoTokenizer.from_pretrained(path, cache_dir=None, model_max_length=model_max_length, padding_side="right",
                                                     use_fast=False, legacy=True)
    tok.pad_token = tok.unk_token
    for name in list(extra_tokens) + [f"<gen_{i}>" for i in range(1, 257)]:
        tok.add_tokens(name, special_tokens=True)
    if use_mm_start_end:
        tok.add_tokens([IM_START, IM_END], special_tokens=True)
    return tok


def from_args(args, cfg, only_val: bool = False):
    """`train.py --dataset medplib_amd.dataset:from_args --data_path x.json --val_data_path y.json --image_folder DIR
    --tokenizer_path DIR [--icl_enable --icl_mask_mode separate --icl_mask_encoder]` -> (train batches, validation batches)."""
    from .collate import collate
    tok = load_tokenizer(args.tokenizer_path, args.model_max_length, extra_tokens=("<SEG>", "<region>", "</region>"))
    dev = torch.device("cuda", torch.cuda.current_device())

    def make(path):
        if getattr(args, "icl_enable", False):
            n_img = cfg.mm_compressed_token_count if getattr(cfg, "mm_token_compress", False) else (cfg.clip_image_size // cfg.clip_patch_size) ** 2
            # is_multimodal=False: the prompt keeps one `<image>` tag per image; the reference's preprocess_multimodal folds the
            # tags of a turn into one (see place_image_token), which leaves its own ICL batches with fewer placeholders than images
            return ICLSupervisedDataset(path, tok, args.image_folder, device=dev, mask_mode=args.icl_mask_mode, is_multimodal=False,
                                        mask_encoder=args.icl_mask_encoder, image_token_len=n_img,
                                        mask_token_len=getattr(cfg, "mask_encoder_token_count", 64), clip_img_size=cfg.clip_image_size)
        return SupervisedDataset(path, tok, args.image_folder, device=dev, clip_img_size=cfg.clip_image_size)

    if only_val:
        val = CollatedBatches(make(args.val_data_path), 1, shuffle=False, collate_fn=lambda s: collate(s, inference=True))
        val.tokenizer = tok
        return val
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    train = CollatedBatches(make(args.data_path), args.batch_size, seed=args.seed, rank=rank, world=world)
    val = CollatedBatches(make(args.val_data_path or args.data_path), 1, shuffle=False)
    return train, val


def val_from_args(args, cfg):
    """`infer.py --dataset medplib_amd.dataset:val_from_args --val_data_path y.json --image_folder DIR --tokenizer_path DIR`:
    single-sample batches through the inference collator (vqa_infer.py:287-314)."""
    return from_args(args, cfg, only_val=True)
