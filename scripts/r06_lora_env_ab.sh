#!/bin/bash
# same-box A/B of the LoRA stage-III step over one environment switch: VAR=<name> VALS="a b" bash scripts/r06_lora_env_ab.sh <tag>
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
tag=${1:-r06w}
for i in 1 2; do
  for v in $VALS; do
    env $VAR=$v python bench.py --lora --steps 12 --warmup 4 --no-kernel-timer 2>gpurun_out/${tag}_${VAR}_$v.err | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$VAR=$v', d['ms_per_step'], 'host issue', d.get('host_issue_ms_per_step'))"
  done
done | tee gpurun_out/${tag}_${VAR}_ab.txt
