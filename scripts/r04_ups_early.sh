#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
UPS_LAB_EARLY=4 ./scripts/lab/ups_lab > gpurun_out/r04_ups_lab_early4.txt 2>&1; sed -n '/timeline/,$p' gpurun_out/r04_ups_lab_early4.txt | head -30
