#!/bin/bash
# same-box sweep of the L2 tile-order group size of the bf16 GEMMs on the whole step
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for g in 4 2 8 16 1 4; do
  echo -n "MP_GEMM_GROUP_M=$g: "
  MP_GEMM_GROUP_M=$g python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-lora-line --no-secondary 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin); print(d['ms_per_step'], d['roofline']['frac'], d['roofline']['avg_launch_us'])"
done
