#!/bin/bash
# Runs every -m gpu test module in its own process (a faulting kernel cannot take the other modules down) and
# collects logs under gpurun_out/.  Usage (on the GPU box, from the repo root): bash scripts/gpu_tests.sh [files...]
mkdir -p gpurun_out/tests
files=("$@")
if [ ${#files[@]} -eq 0 ]; then files=(tests/test_gpu_*.py); fi
rc_all=0
for f in "${files[@]}"; do
  name=$(basename "$f" .py)
  timeout 900 python -m pytest "$f" -q -m gpu --tb=short -s > "gpurun_out/tests/$name.log" 2>&1
  rc=$?
  echo "== $name rc=$rc"
  grep -E "passed|failed|error" "gpurun_out/tests/$name.log" | tail -2
  if [ $rc -ne 0 ]; then rc_all=1; grep -E "max\|err\||Error|error|FAILED|assert" "gpurun_out/tests/$name.log" | head -60; fi
done
exit $rc_all
