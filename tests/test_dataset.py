"""Sample assembly (medplib_amd/dataset.py, host logic) against tests/golden/dataset_reference.json = the reference's own dataset
functions executed by oracle/make_golden.py (`python -m oracle.make_golden dataset`) with tests/toy_tokenizer.py."""
import copy
import json
import os
import random
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from toy_tokenizer import SentencePieceLlamaLike, ToyTokenizer  # noqa: E402

from medplib_amd import dataset as D  # noqa: E402
from oracle.make_golden import dataset_text_cases  # noqa: E402  (the input records; imports nothing from the reference)


@pytest.fixture(scope="module")
def gold(golden_dir):
    return json.load(open(os.path.join(golden_dir, "dataset_reference.json")))


def test_v1_prompt_and_target_masking_match_the_reference(gold):
    tok = ToyTokenizer()
    seen_supervised = 0
    for case, exp in zip(dataset_text_cases(), gold["text"]):
        convs = [copy.deepcopy(case["conversations"])]
        if case["has_image"]:
            D.place_image_token(convs, case["im_start_end"])
        assert convs == exp["placed"], case["name"]
        ex = D.build_v1_example(convs, tok, has_image=case["has_image"])
        assert ex["input_ids"].tolist() == exp["input_ids"], case["name"]
        assert ex["labels"].tolist() == exp["labels"], case["name"]
        assert ex["conversations"] == exp["conversations"] and ex["question"] == exp["question"] and ex["gt"] == exp["gt"]
        seen_supervised += sum(v != D.IGNORE_INDEX for v in exp["labels"][0])
    assert seen_supervised > 20                      # the goldens are not all-IGNORE rows
    by_name = {e["name"]: e for e in gold["text"]}
    assert by_name["several_image_tags_collapse"]["input_ids"][0].count(D.IMAGE_TOKEN_INDEX) == 1
    assert by_name["region_prompt"]["input_ids"][0].count(D.REGION_TOKEN_INDEX) == 1
    assert all(v == D.IGNORE_INDEX for v in by_name["answer_with_separator_text_breaks_round"]["labels"][0])


def test_target_masking_on_a_real_sentencepiece_model(gold, golden_dir):
    """Same cases through a genuine sentencepiece BPE model with the Llama settings: the `- 2` bookkeeping holds (rows are
    supervised, not blanked by the mismatch guard) and ids / labels equal the reference's."""
    tok = SentencePieceLlamaLike(os.path.join(golden_dir, "tiny_llama_like_sp.model"))
    assert tok("USER: hi ASSISTANT: ").input_ids[-1] == tok.sp.piece_to_id("\u2581")          # the lone trailing-space piece
    for case, exp in zip(dataset_text_cases(), gold["text_sp"]):
        convs = [copy.deepcopy(case["conversations"])]
        if case["has_image"]:
            D.place_image_token(convs, case["im_start_end"])
        ex = D.build_v1_example(convs, tok, has_image=case["has_image"])
        assert ex["input_ids"].tolist() == exp["input_ids"] and ex["labels"].tolist() == exp["labels"], case["name"]
        n_sup = sum(v != D.IGNORE_INDEX for v in exp["labels"][0])
        assert (n_sup == 0) == (case["name"] == "answer_with_separator_text_breaks_round"), case["name"]
        if n_sup:       # the supervised ids decode to exactly the assistant turns, each closed by </s>
            sup = [i for i, l in zip(exp["input_ids"][0], exp["labels"][0]) if l != D.IGNORE_INDEX]
            assert sup.count(2) == len(exp["gt"])


def test_generation_stub_prompt():
    p = D.v1_prompt([("USER", "<image>\nhello"), ("ASSISTANT", None)])
    assert p.endswith("USER: <image>\nhello ASSISTANT:") and p.startswith(D.V1_SYSTEM + " ")


def test_mask_and_region_tags(gold, tmp_path):
    from PIL import Image
    g = gold["tags"][0]
    os.makedirs(tmp_path / "m")
    Image.fromarray(np.array([[0, 7, 0], [255, 0, 1]], dtype=np.uint8)).save(tmp_path / "m" / "a_mask.png")
    Image.fromarray(np.array([[0, 0], [3, 0]], dtype=np.uint8)).save(tmp_path / "r.png")
    rec = copy.deepcopy(g["source"])
    names = D.pull_tagged_files(rec, "mask")
    rnames = D.pull_tagged_files(rec, "region")
    assert rec == g["after"] and names == ["m/a_mask.png"] and rnames == ["r.png"]
    ds = D.SupervisedDataset([], ToyTokenizer(), str(tmp_path), device="cpu")
    assert [ds._load_binary(n).tolist() for n in names] == g["masks"]
    assert [ds._load_binary(n).tolist() for n in rnames] == g["regions"]
    with pytest.raises(AssertionError):
        D.pull_tagged_files({"conversations": [{"from": "gpt", "value": "no seg <mask>a.png</mask>"}]}, "mask")


def test_region_subcomponents_consume_the_same_random_stream(gold):
    for g in gold["subregion"]:
        random.seed(g["seed"])
        subs, ok = D.region_subcomponents([np.array(m, dtype=np.float32) for m in g["masks"]], min_area=0.2, max_area=1, min_thresh=10)
        assert ok == g["valid"], g["seed"]
        assert [np.asarray(s).astype(int).tolist() for s in subs] == g["subs"], g["seed"]
        assert random.random() == g["next_draw"], g["seed"]          # same number of draws, in the same order


def test_icl_record_helpers(gold):
    for g in gold["icl"]:
        raw = copy.deepcopy(g["record"])
        ex = D.icl_examples_of(raw)
        assert ex == g["examples"] and raw == g["raw_after"]
        assert D.icl_prepare_source(raw, len(ex), g["mode"]) == g["prepared"]
    o = gold["overlay"]
    out = D.overlay_mask(np.array(o["image"], dtype=np.uint8), np.array(o["mask"], dtype=np.uint8))
    assert out.dtype == np.uint8 and out.tolist() == o["out"]


def test_collated_batches_shard_disjointly_across_ranks():
    samples = list(range(23))
    views = [D.CollatedBatches(samples, 3, seed=5, rank=r, world=2, collate_fn=lambda s: s) for r in range(2)]
    assert len(views[0]) == 4
    seen = []
    for i in range(3):                       # 3 strides of 2 x 3 = 18 of the 23 samples, no repeats across ranks / steps
        a, b = views[0][i], views[1][i]
        assert len(a) == len(b) == 3 and not set(a) & set(b)
        seen += a + b
    assert len(set(seen)) == 18
    assert views[0][3] != views[0][0] and len(views[0][3]) == 3          # wraps into the next epoch's permutation
    again = D.CollatedBatches(samples, 3, seed=5, rank=0, world=2, collate_fn=lambda s: s)
    assert [again[i] for i in range(4)] == [views[0][i] for i in range(4)]          # same order in every run
    flat = D.CollatedBatches(samples, 4, shuffle=False, collate_fn=lambda s: s)
    assert flat[0] == [0, 1, 2, 3] and flat[5] == [20, 21, 22, 0]


def test_target_masking_invariants_on_random_conversations():
    """Property test (hypothesis): for any alternating conversation the supervised positions carry exactly the assistant turns,
    each closed by </s>, in order; everything else (system prompt, user turns, image sentinel, padding) is IGNORE; ids and labels
    agree wherever a label is kept."""
    from hypothesis import given, settings, strategies as st
    tok = ToyTokenizer()
    word = st.text(alphabet="abcdefghijklmnopqrstuvwxyz.,?", min_size=1, max_size=7)
    sentence = st.lists(word, min_size=1, max_size=6).map(" ".join)
    rounds = st.lists(st.tuples(sentence, sentence), min_size=1, max_size=4)

    @settings(max_examples=60, deadline=None)
    @given(rounds=rounds, with_image=st.booleans(), seg=st.booleans())
    def check(rounds, with_image, seg):
        conv = []
        for k, (q, a) in enumerate(rounds):
            conv.append({"from": "human", "value": (q + " <image>" if (with_image and k == 0) else q)})
            conv.append({"from": "gpt", "value": a + (" <SEG>" if seg and k == len(rounds) - 1 else "")})
        convs = [copy.deepcopy(conv)]
        if with_image:
            D.place_image_token(convs)
            assert convs[0][0]["value"].startswith("<image>\n")
        ex = D.build_v1_example(convs, tok, has_image=with_image)
        ids, lab = ex["input_ids"][0].tolist(), ex["labels"][0].tolist()
        assert len(ids) == len(lab) and ids.count(D.IMAGE_TOKEN_INDEX) == (1 if with_image else 0)
        assert all(l == i for i, l in zip(ids, lab) if l != D.IGNORE_INDEX)
        kept = [l for l in lab if l != D.IGNORE_INDEX]
        want = []
        for t in convs[0][1::2]:                      # " answer</s>" of every assistant turn: the space merges into the first word
            want += tok(" " + t["value"], add_special_tokens=False).input_ids[1:] + [2]
        assert kept == want
        assert lab[0] == D.IGNORE_INDEX and ex["gt"] == [t["value"] for t in convs[0][1::2]]

    check()
