#!/bin/bash
# same-box A/B: the run-ahead towers' layers gated on the previous step's decoder layers (MP_GATE_TOWERS=1) against free-running side streams
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
tag=${1:-r06n}
F="--steps 20 --warmup 5 --no-cpu-baseline --no-lora-line --no-secondary --no-live-traffic --roofline-steps 0"
for i in 1 2; do
  for v in ${MODES:-0 1}; do
    MP_GATE_TOWERS=$v python bench.py $F 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('MP_GATE_TOWERS=$v', d['ms_per_step'], d['value'], d.get('host_issue_ms_per_step'))"
  done
done | tee gpurun_out/${tag}_gate_ab.txt
