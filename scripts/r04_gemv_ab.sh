#!/bin/bash
# same-box A/B of a GEMV change in the decode step: shipped library vs medplib_amd/lib/ab/libmedplib_hip_old.so (the previous commit's gemv_bf16.hip)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
python -m pytest tests -m gpu -x -q -k "decode or evaluate or gemv" 2>&1 | tail -3
for rep in 1 2; do
for lib in "" medplib_amd/lib/ab/libmedplib_hip_old.so; do
  echo -n "lib=${lib:-shipped}: "
  MEDPLIB_HIP_LIB=${lib:+$GRAFT_REPO_ROOT/$lib} python scripts/decode_bench.py 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); print(d['value'], d['frac_of_8TBps'])"
  echo -n "   dense: "
  MEDPLIB_HIP_LIB=${lib:+$GRAFT_REPO_ROOT/$lib} python scripts/decode_bench.py --dense 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); print(d['value'], d['frac_of_8TBps'])"
done; done
