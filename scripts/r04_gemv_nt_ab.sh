#!/bin/bash
# same-box A/B of non-temporal weight loads in the decode GEMVs (build -DMP_GEMV_NT=0 = default cache policy)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for rep in 1 2; do
for lib in "" medplib_amd/lib/ab/libmedplib_hip_nt0.so; do
  echo -n "lib=${lib:-shipped (nt)}: "
  MEDPLIB_HIP_LIB=${lib:+$GRAFT_REPO_ROOT/$lib} python scripts/decode_bench.py 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); print(d['value'], d['frac_of_8TBps'])"
  echo -n "   dense: "
  MEDPLIB_HIP_LIB=${lib:+$GRAFT_REPO_ROOT/$lib} python scripts/decode_bench.py --dense 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); print(d['value'], d['frac_of_8TBps'])"
done; done
python -m pytest tests -m gpu -x -q -k "decode or evaluate or gemv" 2>&1 | tail -3
