"""Host-side index planning for the multimodal splice and the <SEG> mask (integer work on [B, L] token ids).

The reference does this with per-sample / per-token Python loops over GPU tensors
(`prepare_inputs_labels_for_multimodal`, model/medplib/model/medplib_arch.py:217-527;
`build_seg_token_mask`, model/MedPLIB.py:310-355).  Here the plan is computed once per batch with numpy on the host
(B*L integers) and the data movement is a single gather kernel (mp_splice_rows_bf16)."""
from dataclasses import dataclass
from typing import List, Optional, Sequence

import numpy as np

IGNORE_INDEX = -100          # utils/utils.py:7
IMAGE_TOKEN_INDEX = -200     # utils/utils.py:8
REGION_TOKEN_INDEX = -300    # utils/utils.py:9
SPLICE_PAD = -(2 ** 63)


@dataclass
class SplicePlan:
    src_code: np.ndarray      # int64 [B, S]: >=0 token id, SPLICE_PAD pad row, else -1 - feature_row
    labels: Optional[np.ndarray]   # int64 [B, S] or None
    attention_mask: Optional[np.ndarray]  # bool [B, S] or None
    seg_mask: Optional[np.ndarray]  # bool [B, S] or None
    lengths: np.ndarray       # int64 [B] un-padded spliced lengths
    n_feature_rows: int

    @property
    def seq_len(self):
        return self.src_code.shape[1]

    def supervised(self):
        """(flat row ids, labels) of every position whose NEXT-token label is not IGNORE (shifted CE,
        medplib_moe_llama.py:394-408; whole rows without supervision contribute nothing, :399-402)."""
        lab = self.labels[:, 1:]
        b, t = np.nonzero(lab != IGNORE_INDEX)
        return (b * self.seq_len + t).astype(np.int64), lab[b, t].astype(np.int64)

    def seg_rows(self):
        """flat row ids of the <SEG> positions in boolean-indexing (row-major) order (MedPLIB.py:461)."""
        return np.flatnonzero(self.seg_mask.reshape(-1)).astype(np.int64)


def plan_splice(input_ids: np.ndarray, labels: Optional[np.ndarray], attention_mask: Optional[np.ndarray],
                feature_lengths, images_per_sample: Optional[Sequence[int]] = None, seg_token_idx: Optional[int] = None,
                seg_feature_lengths=None, feature_bases: Optional[Sequence[int]] = None,
                region_bases: Optional[Sequence[Optional[int]]] = None) -> SplicePlan:
    """input_ids [B, L] int64 with IMAGE_TOKEN_INDEX placeholders.

    feature_lengths: int (every image expands to that many rows; one image per sample, consumed in batch order even by
      samples without a placeholder — medplib_arch.py:299-313) or a flat list with one entry per placeholder in
      (sample, position) order (multi-image ICL layouts, medplib_arch.py:246-278).
    seg_feature_lengths: the lengths build_seg_token_mask uses (image_token_len or image_token_lengths[b][k],
      MedPLIB.py:318-341); defaults to feature_lengths.
    feature_bases: first feature row of each placeholder (flat, with per-placeholder feature_lengths) when the feature
      buffer is not laid out in placeholder order — ICL separate mode keeps all image blocks first and the mask-encoder
      blocks after them (medplib_arch.py:246-267 interleaves them by `image_token_types`).
    region_bases: per sample, the feature row of its first region feature (or None): the k-th REGION_TOKEN_INDEX after the
      sample's last image placeholder is replaced by row region_bases[b] + k (medplib_arch.py:409-433; labels keep those
      positions).  Region tokens anywhere else are rejected, like the reference's assert (:323-324)."""
    ids = np.asarray(input_ids, dtype=np.int64)
    B, L = ids.shape
    assert region_bases is not None or not (ids == REGION_TOKEN_INDEX).any(), "REGION_TOKEN_INDEX ids need region features"
    per_token = not np.isscalar(feature_lengths)
    assert feature_bases is None or per_token, "feature_bases needs per-placeholder feature_lengths"
    flat_lengths = list(feature_lengths) if per_token else None
    seg_lens = seg_feature_lengths if seg_feature_lengths is not None else feature_lengths
    seg_per_sample = (not np.isscalar(seg_lens)) and len(seg_lens) > 0 and not np.isscalar(seg_lens[0])
    rows_src, rows_lab, rows_seg, lens = [], [], [], []
    feat_row, feat_idx = 0, 0
    seg_shift = None
    if seg_token_idx is not None:
        seg_shift = np.zeros_like(ids, dtype=bool)
        seg_shift[:, :-1] = ids[:, 1:] == seg_token_idx
    for b in range(B):
        cur = ids[b]
        pos = np.flatnonzero(cur == IMAGE_TOKEN_INDEX)
        src_parts, lab_parts, seg_parts = [], [], []
        prev = 0
        if pos.size == 0 and not per_token:
            nfeat = int(feature_lengths)
            feat_row += nfeat            # the sample's (unused) image still occupies its feature rows
        for k, p in enumerate(pos):
            nfeat = int(flat_lengths[feat_idx]) if per_token else int(feature_lengths)
            base = feat_row if feature_bases is None else int(feature_bases[feat_idx])
            src_parts.append(cur[prev:p])
            src_parts.append(-1 - (base + np.arange(nfeat, dtype=np.int64)))
            if labels is not None:
                lab_parts.append(labels[b, prev:p])
                lab_parts.append(np.full(nfeat, IGNORE_INDEX, dtype=np.int64))
            if seg_shift is not None:
                if np.isscalar(seg_lens):
                    nseg = int(seg_lens)
                elif seg_per_sample:
                    nseg = int(seg_lens[b][k]) if (len(seg_lens) > b and len(seg_lens[b]) > k) else nfeat
                else:
                    nseg = int(seg_lens[feat_idx])
                seg_parts.append(seg_shift[b, prev:p])
                seg_parts.append(np.zeros(nseg, dtype=bool))
            feat_row += nfeat
            feat_idx += 1
            prev = p + 1
            if not per_token and k == 0 and pos.size > 1:
                raise ValueError("several image placeholders in one sample need per-placeholder feature_lengths")
        tail = cur[prev:]
        rpos = np.flatnonzero(tail == REGION_TOKEN_INDEX)
        assert not (cur[:prev] == REGION_TOKEN_INDEX).any(), "region tokens must follow the sample's last image placeholder"
        if rpos.size:
            assert pos.size > 0 and region_bases is not None and region_bases[b] is not None, \
                "REGION_TOKEN_INDEX in a sample without image / region features"
            tail = tail.copy()
            tail[rpos] = -1 - (int(region_bases[b]) + np.arange(rpos.size, dtype=np.int64))
        src_parts.append(tail)
        if labels is not None:
            lab_parts.append(labels[b, prev:])
        if seg_shift is not None:
            seg_parts.append(seg_shift[b, prev:])
        rows_src.append(np.concatenate(src_parts))
        lens.append(rows_src[-1].shape[0])
        if labels is not None:
            rows_lab.append(np.concatenate(lab_parts))
        if seg_shift is not None:
            rows_seg.append(np.concatenate(seg_parts))
    S = max(lens)
    src = np.full((B, S), SPLICE_PAD, dtype=np.int64)
    lab = np.full((B, S), IGNORE_INDEX, dtype=np.int64) if labels is not None else None
    att = np.zeros((B, S), dtype=bool) if attention_mask is not None else None
    for b in range(B):
        n = lens[b]
        src[b, :n] = rows_src[b]
        if lab is not None:
            lab[b, :n] = rows_lab[b]
        if att is not None:
            # left-extend with True by the added length, keep the original mask, right-pad False (medplib_arch.py:480-526)
            add = n - L
            att[b, :add] = True
            att[b, add:n] = np.asarray(attention_mask[b], dtype=bool)
    seg = None
    if seg_shift is not None:
        Sg = max(r.shape[0] for r in rows_seg)
        seg = np.zeros((B, Sg), dtype=bool)
        for b in range(B):
            seg[b, :rows_seg[b].shape[0]] = rows_seg[b]
    return SplicePlan(src, lab, att, seg, np.asarray(lens, dtype=np.int64), feat_row)


def icl_feature_layout(image_token_types, image_len: int, mask_len: int):
    """Per-placeholder (lengths, bases) for ICL separate mode: the feature buffer holds every image block (image_len rows each,
    in image order) followed by every mask-encoder block (mask_len rows each, in mask order); placeholders draw from the two
    lists in the order `image_token_types` gives (medplib_arch.py:256-266)."""
    n_img = sum(1 for row in image_token_types for t in row if t != "mask")
    lengths, bases = [], []
    ii = mi = 0
    for row in image_token_types:
        for t in row:
            if t == "mask":
                lengths.append(mask_len); bases.append(n_img * image_len + mi * mask_len); mi += 1
            else:
                lengths.append(image_len); bases.append(ii * image_len); ii += 1
    return lengths, bases
