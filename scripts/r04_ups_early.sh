#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
for tf in 0 1 0 1; do echo "== UPS_LAB_TF=$tf (early 4)"; UPS_LAB_TF=$tf ./scripts/lab/ups_lab > gpurun_out/r04_ups_lab_tf$tf.txt 2>&1; grep -E "^both: full kernel|graph of|tokens landed|after barrier|GEMM1 issued|stores drained" gpurun_out/r04_ups_lab_tf$tf.txt | head -7; done
grep "    wave" gpurun_out/r04_ups_lab_tf1.txt | head -16
