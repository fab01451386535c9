#!/bin/bash
# the fused SAM encoder: its tests, the tower alone, what the towers cost the step
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
tag=${1:-r06b}
timeout 900 python -m pytest tests/test_gpu_sam_fused.py -q -m gpu -x -s 2>&1 | tail -40
timeout 600 python -m pytest tests/test_gpu_model.py -q -m gpu -x -k "sam_encoder or lisa or model_forward_golden" 2>&1 | tail -5
for t in sam; do
  rm -rf gpurun_out/prof_$t
  TOWER=$t timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_$t -- python scripts/r06_tower_trace.py 10 > gpurun_out/${tag}_${t}_alone.log 2>&1
  db=$(ls gpurun_out/prof_$t/*/*.db | head -1)
  python scripts/rocpd_stats.py $db 13 gpurun_out/${tag}_${t}_alone_kernel_stats.md > /dev/null
  (cd scripts && python rocpd_cutime.py ../$db 13 ../gpurun_out/${tag}_${t}_alone_cu_time.md > /dev/null)
  grep "ms per forward" gpurun_out/${tag}_${t}_alone.log
  rm -rf gpurun_out/prof_$t
done
timeout 900 python scripts/tower_cost.py > gpurun_out/${tag}_tower_cost.txt 2>&1; tail -5 gpurun_out/${tag}_tower_cost.txt
MP_SAM_FUSED=0 timeout 900 python scripts/tower_cost.py > gpurun_out/${tag}_tower_cost_generic.txt 2>&1; tail -5 gpurun_out/${tag}_tower_cost_generic.txt
