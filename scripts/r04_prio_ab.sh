#!/bin/bash
# same-box A/B of the 320-row GEMM's priority scheme around its MFMA segments (builds: -DMP3_PRIO=0/1/2, see gemm320_bf16.hip)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for rep in 1 2; do
for lib in "" medplib_amd/lib/ab/libmedplib_hip_prio1.so medplib_amd/lib/ab/libmedplib_hip_prio2.so; do
  echo -n "lib=${lib:-shipped}: "
  MEDPLIB_HIP_LIB=${lib:+$GRAFT_REPO_ROOT/$lib} python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-lora-line --no-secondary 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin); print(d['ms_per_step'], d['roofline']['frac'], d['roofline']['avg_launch_us'], d['loss_last'])"
done; done
