#!/bin/bash
# round 4: one upsampler iteration on the GPU box — lab timings + timeline, then the kernel's parity tests
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
tag=${1:-a}
./scripts/lab/ups_lab > gpurun_out/r04_ups_lab_$tag.txt 2>&1; cat gpurun_out/r04_ups_lab_$tag.txt | head -16
