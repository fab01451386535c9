#!/bin/bash
# folded input norms: kernel / stack tests, the 8-layer B = 8 parity with the fold on, same-box A/B of the headline step
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
tag=${1:-r06k}
timeout 900 python -m pytest tests/test_gpu_fold_norm.py -q -m gpu -x -s 2>&1 | grep -v "^$" | tail -12
timeout 1200 python -m pytest tests/test_gpu_model.py -q -m gpu -x -s -k "batch8" 2>&1 | grep -E "passed|failed|Error|assert|folded|rows_agreeing" | cut -c1-400 | tail -12
F="--steps 20 --warmup 5 --no-cpu-baseline --no-lora-line --no-secondary --no-live-traffic --roofline-steps 0"
for i in 1 2; do
  for v in "" "--fold-input-norm"; do
    python bench.py $F $v 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('fold=$v', d['ms_per_step'], d['config'].get('folded_norm_layers'))"
  done
done | tee gpurun_out/${tag}_fold_ab.txt
