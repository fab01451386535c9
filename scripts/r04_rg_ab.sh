#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
sed -i 's#sys.path.insert(0, "/root/repo")#import os; sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))#' scripts/rmsnorm_gate_bench.py
for w in 4 8 4 8; do echo -n "MP_RG_WAVES=$w: "; MP_RG_WAVES=$w python scripts/rmsnorm_gate_bench.py 2>&1 | tail -1; done
python -m pytest tests/test_gpu_trunk_kernels.py -x -q -k "rmsnorm" 2>&1 | tail -3
python -m pytest tests/test_gpu_model.py -x -q -k "moe_gather_scatter or rmsnorm_gate or moe_layer" 2>&1 | tail -3
