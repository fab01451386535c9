#!/bin/bash
# decode: K-split narrow projections (MP_GEMV_KSPLIT=1, default) vs one wave per four rows (0), same box
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
python -m pytest tests/test_gpu_trunk_kernels.py -x -q -k "gemv" 2>&1 | tail -3
for rep in 1 2; do for ks in 1 0; do
  echo -n "MP_GEMV_KSPLIT=$ks: "; MP_GEMV_KSPLIT=$ks python scripts/decode_bench.py 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); print(d['value'], d['frac_of_8TBps'])"
  echo -n "   dense: "; MP_GEMV_KSPLIT=$ks python scripts/decode_bench.py --dense 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); print(d['value'], d['frac_of_8TBps'])"
done; done
python -m pytest tests -m gpu -x -q -k "decode or evaluate or gemv" 2>&1 | tail -3
