#!/bin/bash
# round 3: the biting parity checks (evaluate golden, LISA golden mask cuts, full-depth B=1 / B=8 RTS) + the default bench line
mkdir -p gpurun_out/r03a
timeout 1500 python -m pytest tests/test_gpu_model.py -q -m gpu --tb=short -s -k "evaluate_vs_executed or executed_reference_lisa or full_depth_parity or icl_separate_mode_parity" > gpurun_out/r03a/parity_tests.log 2>&1
echo "parity tests rc=$?"; grep -E "passed|failed" gpurun_out/r03a/parity_tests.log | tail -2
grep -E "flipped|generated|AssertionError|assert " gpurun_out/r03a/parity_tests.log | head -60
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r03a/bench.json 2> gpurun_out/r03a/bench.err
echo "bench rc=$?"; tail -3 gpurun_out/r03a/bench.err; python - <<'P'
import json
try:
    d = json.load(open("gpurun_out/r03a/bench.json"))
    print(d["value"], d["ms_per_step"], json.dumps(d.get("parity"))[:3000])
except Exception as e:
    print("no bench json", e)
P
