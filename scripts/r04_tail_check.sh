#!/bin/bash
# round 4, first contact: the bounded-wait split tail (tests + cost of the fallback path) and a baseline bench line at HEAD
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
python -m pytest tests/test_gpu_trunk_kernels.py -k "gemm320 or gemm_320" -x -q 2>&1 | tail -8
echo "== expert gemm A/B, default wait"; python scripts/expert_gemm_ab.py 2>&1 | grep down
echo "== expert gemm A/B, wait 0 (every tile through the fallback decision)"; MP_GEMM320_TAIL_WAIT=0 python scripts/expert_gemm_ab.py 2>&1 | grep down
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r04a_bench.json 2> gpurun_out/r04a_bench.err
echo "== bench rc=$?"; python -c "
import json; d=json.load(open('gpurun_out/r04a_bench.json')); print(d['ms_per_step'], d['value'], d['roofline']['frac'], d['lora_stage3']['ms_per_step'])"
