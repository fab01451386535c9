#!/bin/bash
# same-box A/B of the headline step over one environment switch: VAR=<name> VALS="a b" bash scripts/r06_env_ab.sh <tag>
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
tag=${1:-r06q}
F="--steps 20 --warmup 5 --no-cpu-baseline --no-lora-line --no-secondary --no-live-traffic --roofline-steps 0"
for i in 1 2; do
  for v in $VALS; do
    env $VAR=$v python bench.py $F 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$VAR=$v', d['ms_per_step'], d['value'])"
  done
done | tee gpurun_out/${tag}_${VAR}_ab.txt
