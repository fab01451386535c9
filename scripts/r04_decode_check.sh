#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
python -m pytest tests/test_gpu_trunk_kernels.py -x -q -k "attention or decode" 2>&1 | tail -3
python -m pytest tests/test_gpu_model.py -x -q -k "evaluate" 2>&1 | tail -3
python scripts/decode_bench.py --dense 2>&1 | tail -1
python scripts/decode_bench.py 2>&1 | tail -1
