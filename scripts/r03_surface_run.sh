#!/bin/bash
mkdir -p gpurun_out/r03b
timeout 1200 python -m pytest tests/test_gpu_surface.py -q -m gpu --tb=short -s > gpurun_out/r03b/surface.log 2>&1
echo "surface rc=$?"; grep -E "passed|failed" gpurun_out/r03b/surface.log | tail -2; grep -E "Error|error|assert" gpurun_out/r03b/surface.log | head -40
timeout 1500 python -m pytest tests/test_gpu_model.py -q -m gpu --tb=short -s -k "evaluate_vs_executed or executed_reference_lisa or full_depth_parity or icl_separate_mode_parity or greedy" > gpurun_out/r03b/parity_tests.log 2>&1
echo "parity tests rc=$?"; grep -E "passed|failed" gpurun_out/r03b/parity_tests.log | tail -2
grep -E "AssertionError|assert |^E " gpurun_out/r03b/parity_tests.log | head -40
