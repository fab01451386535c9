#!/bin/bash
# LoRA stage-III step: the adapter-branch switches of round 5 (MP_LORA_FUSE_SWIGLU_SKINNY, MP_LORA_FUSE_NORM_UP, MP_LORA_FUSE_DY, MP_LORA_KEEP_BITS) and the
# pruned last layer (MP_PRUNE_LAST_MLP): tests, the newest switch off / on twice, then everything of the round off — same box
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_lora_fused.py tests/test_gpu_prune_last_mlp.py -x -q -m gpu 2>&1 | tail -4
timeout 1500 python -m pytest tests/test_gpu_model.py -x -q -m gpu -k "lora" 2>&1 | grep -E "passed|failed|Error" | tail -4
run() { tag=$1; shift
  env "$@" python bench.py --lora --steps 12 --warmup 3 --no-kernel-timer --no-cpu-baseline --no-secondary --no-live-traffic 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$tag', d['ms_per_step'], d.get('host_issue_ms_per_step'))"
}
run sw_off MP_LORA_FUSE_SWIGLU_SKINNY=0; run sw_on; run sw_off2 MP_LORA_FUSE_SWIGLU_SKINNY=0; run sw_on2
run round_start MP_LORA_FUSE_SWIGLU_SKINNY=0 MP_LORA_FUSE_NORM_UP=0 MP_LORA_FUSE_DY=0 MP_LORA_KEEP_BITS=0 MP_PRUNE_LAST_MLP=0
