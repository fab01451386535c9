"""CPU: the import faces the reference's drivers use besides `model.*` — `datasets`, `utils.utils` (train_ds_medplib.py:23-26,
model/eval/vqa_infer.py:26-29) — and the two surface walks' flag tables."""
import io
import os
import sys
from contextlib import redirect_stdout

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def test_driver_import_lines_resolve():
    from datasets import DataCollatorForSupervisedDataset, ICLLazySupervisedDataset, LazySupervisedDataset
    from utils.utils import (ADD_OTHERS_TOKENS, DEFAULT_IM_END_TOKEN, DEFAULT_IM_START_TOKEN, AverageMeter, ProgressMeter, Summary,
                             dict_to_cuda, intersectionAndUnionGPU)
    import medplib_amd.collate as C
    import medplib_amd.dataset as D
    assert DataCollatorForSupervisedDataset is C.collate and LazySupervisedDataset is D.LazySupervisedDataset
    assert ICLLazySupervisedDataset is D.ICLLazySupervisedDataset
    assert (DEFAULT_IM_START_TOKEN, DEFAULT_IM_END_TOKEN) == ("<im_start>", "<im_end>")
    assert ADD_OTHERS_TOKENS[0] == "<SEG>" and "<region>" in ADD_OTHERS_TOKENS and "<mask>" in ADD_OTHERS_TOKENS and len(ADD_OTHERS_TOKENS) == 9
    assert callable(dict_to_cuda) and callable(intersectionAndUnionGPU) and Summary.SUM.value == 2 and AverageMeter and ProgressMeter


def test_average_meter_and_progress_meter_formats():
    from utils.utils import AverageMeter, ProgressMeter, Summary
    m = AverageMeter("Loss", ":.4f")
    m.update(2.0, 2); m.update(5.0, 1)
    assert (m.val, m.sum, m.count, m.avg) == (5.0, 9.0, 3, 3.0)
    assert str(m) == "Loss 5.0000 (3.0000)" and m.summary() == "Loss 3.000"
    s = AverageMeter("Intersec", ":6.3f", Summary.SUM)
    s.update(np.array([1.0, 2.0])); s.update(np.array([3.0, 4.0]))
    assert np.array_equal(s.sum, [4.0, 6.0]) and s.count == 2
    s.all_reduce()                                     # no process group: the sums stay, avg = sum / (count + 1e-5)
    assert np.allclose(s.sum, [4.0, 6.0]) and np.allclose(s.avg, np.array([4.0, 6.0]) / (2 + 1e-5))
    assert AverageMeter("x", summary_type=Summary.NONE).summary() == ""
    buf = io.StringIO()
    with redirect_stdout(buf):
        p = ProgressMeter(120, [m], prefix="Epoch: [3]")
        p.display(7)
        p.display_summary()
    assert buf.getvalue().splitlines() == ["Epoch: [3][  7/120]\tLoss 5.0000 (3.0000)", " * Loss 3.000"]


def test_intersection_and_union_counts():
    """Against the definition (three histograms over K bins, utils/utils.py:92-104), incl. ignored pixels and the in-place write."""
    from utils.utils import intersectionAndUnionGPU
    g = torch.Generator().manual_seed(0)
    for K in (2, 5):
        out = torch.randint(0, K, (3, 17, 19), generator=g)
        tgt = torch.randint(0, K, (3, 17, 19), generator=g)
        tgt[torch.rand(tgt.shape, generator=g) < 0.1] = 255
        o2 = out.clone()
        i, u, t = intersectionAndUnionGPU(o2, tgt, K, ignore_index=255)
        ref_o = out.clone().view(-1); ref_t = tgt.view(-1)
        ref_o[ref_t == 255] = 255
        hist = lambda x: torch.histc(x.float(), bins=K, min=0, max=K - 1)
        ri = hist(ref_o[ref_o == ref_t]); ro = hist(ref_o); rt = hist(ref_t)
        assert torch.equal(i, ri) and torch.equal(u, ro + rt - ri) and torch.equal(t, rt)
        assert torch.equal(o2.view(-1), ref_o)          # the documented in-place effect on `output`


def test_surface_walk_flag_tables():
    import ast
    for rel, must in (("train_ds_medplib.py", {"version", "sft_modules", "lora_r", "moe_enable", "num_experts", "ep_size", "auto_resume",
                                                "grad_accumulation_steps", "load_in_8bit", "exclude_val"}),
                      ("model/eval/vqa_infer.py", {"version", "eval_seg", "eval_vqa", "temperature", "num_beams", "answers-file",
                                                   "num-chunks", "return_gating_logit"})):
        tree = ast.parse(open(os.path.join(ROOT, rel)).read())
        table = next(n.value for n in tree.body if isinstance(n, ast.Assign) and getattr(n.targets[0], "id", "") == "FLAG_TABLE")
        names = [e.elts[0].value for e in table.elts]
        assert len(names) == len(set(names)) and must <= set(names), (rel, must - set(names))


def test_surface_walk_command_lines_parse_like_the_reference():
    """The walks accept the reference's command lines: scripts/train_stage4.sh's flag set (argparse `type=bool` reads any non-empty string
    as True, which `--moe_enable True` relies on), store_true switches, choices; engine_config builds the ds_config of :383-420 from them."""
    import train_ds_medplib as T
    a = T.parse_args(["--version", "/ckpt", "--moe_enable", "True", "--moe_mode", "dense", "--num_experts", "2", "--top_k_experts", "1",
                      "--capacity_factor", "1.5", "--lora_r", "8", "--lora_target_modules", "gate_proj,up_proj,down_proj,q_proj,v_proj",
                      "--sft_modules", "wg,lm_head,embed_tokens,mask_decoder,text_hidden_fcs", "--train_mask_decoder", "--epochs", "3",
                      "--batch_size", "8", "--grad_accumulation_steps", "2", "--lr", "2e-4", "--no_eval", "--icl_mask_mode", "separate"])
    assert a.moe_enable is True and a.num_experts == 2 and a.top_k_experts == 1 and a.capacity_factor == 1.5 and a.train_mask_decoder
    assert a.gradient_checkpointing and a.use_mm_start_end and a.auto_resume and a.no_eval and not a.eval_only      # defaults kept
    assert a.router_aux_loss_coef == 0.01 and a.eval_capacity_factor == 2.0 and a.ep_size == 1 and a.precision == "bf16"
    a.steps_per_epoch = 500
    c = T.engine_config(a, a.batch_size)
    assert c["train_micro_batch_size_per_gpu"] == 8 and c["gradient_accumulation_steps"] == 2 and c["gradient_clipping"] == 1.0
    assert c["optimizer"]["params"] == {"lr": 2e-4, "weight_decay": 0.0, "betas": (0.9, 0.95)} and c["bf16"]["enabled"] and not c["fp16"]["enabled"]
    assert c["scheduler"]["params"] == {"total_num_steps": 1500, "warmup_min_lr": 0, "warmup_max_lr": 2e-4, "warmup_num_steps": 5,
                                        "warmup_type": "linear"}
    assert c["zero_optimization"]["stage"] == 2 and c["zero_optimization"]["overlap_comm"]
    assert [s.__name__ for s, _ in T.STAGES][:3] == ["stage_process_setup", "stage_tokenizer", "stage_open_checkpoint"] and len(T.STAGES) == 11
    from model.eval import vqa_infer as V
    v = V.parse_args(["--version", "/ckpt", "--eval_seg", "--moe_enable", "--num-chunks", "4", "--chunk-idx", "1", "--answers-file", "/tmp/a.jsonl",
                      "--temperature", "0", "--num_beams", "1"])
    assert v.eval_seg and not v.eval_vqa and v.moe_enable and v.num_chunks == 4 and v.chunk_idx == 1 and v.answers_file == "/tmp/a.jsonl"
    assert v.model_max_length == 2048 and v.is_multimodal and v.use_mm_start_end and v.val_batch_size == 1
