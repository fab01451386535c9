#!/bin/bash
# same-box A/B: workgroups of the mask tail's persistent program while it runs hidden beside the next step's decoder (headline step)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
tag=${1:-r06h}
F="--steps 20 --warmup 5 --no-cpu-baseline --no-lora-line --no-secondary --no-live-traffic --roofline-steps 0"
for i in 1 2; do
  for v in ${GRIDS:-64 32 16 8}; do
    MP_TAIL_GRID=$v python bench.py $F 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('MP_TAIL_GRID=$v', d['ms_per_step'], d.get('dp_bucket',{}).get('tail_backward_us'))"
  done
done | tee gpurun_out/${tag}_tailgrid_ab.txt
