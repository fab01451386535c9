#!/bin/bash
# round 4: one upsampler iteration on the GPU box — lab timings + timeline, then the kernel's parity tests
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
tag=${1:-a}
./scripts/lab/ups_lab > gpurun_out/r04_ups_lab_$tag.txt 2>&1; grep -v "    wave" gpurun_out/r04_ups_lab_$tag.txt | head -16
python -m pytest tests/test_gpu_mask_tail_kernels.py -x -q 2>&1 | tail -3
python -m pytest tests/test_gpu_model.py -x -q -k "fused or golden" 2>&1 | tail -3
python scripts/upsampler_bench.py 2>&1 | tail -2
