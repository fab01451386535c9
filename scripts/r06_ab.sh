#!/bin/bash
# same-box A/B of the headline step: the SAM encoder's own kernels (default) against the generic launches (MP_SAM_FUSED=0), twice each, alternating
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
tag=${1:-r06f}
F="--steps 20 --warmup 5 --no-cpu-baseline --no-lora-line --no-secondary --no-live-traffic --roofline-steps 0"
for i in 1 2; do
  for v in 1 0; do
    MP_SAM_FUSED=$v python bench.py $F 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('MP_SAM_FUSED=$v', d['ms_per_step'], d['value'])"
  done
done | tee gpurun_out/${tag}_sam_ab.txt
