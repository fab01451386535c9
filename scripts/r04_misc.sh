#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
python -m pytest tests/test_gpu_model.py -x -q -k "gemm_timer or gather_scatter" 2>&1 | tail -6
