#!/bin/bash
# end-of-round validation on the GPU box: every -m gpu module, smoke(), the driver's bench command (timed), profiles at HEAD
tag=${1:-r03b}
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
bash scripts/gpu_tests.sh; echo "== tests rc=$?"
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
t0=$(date +%s)
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err
echo "== bench rc=$? wall $(( $(date +%s) - t0 )) s, stdout lines: $(wc -l < gpurun_out/${tag}_bench.json)"; grep "gpu leg" gpurun_out/${tag}_bench.err
rm -rf gpurun_out/prof_$tag
timeout 900 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_$tag -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-lora-line > /dev/null 2> gpurun_out/${tag}_bench_prof.err
db=$(ls gpurun_out/prof_$tag/*/*.db | head -1); python scripts/rocpd_stats.py $db 11 gpurun_out/${tag}_kernel_stats.md; head -14 gpurun_out/${tag}_kernel_stats.md
rm -rf gpurun_out/prof_$tag
TAG=$tag bash scripts/bench_pmc.sh > gpurun_out/${tag}_pmc.log 2>&1; tail -3 gpurun_out/${tag}_pmc.log
TAG=$tag bash scripts/bench_mfma_pmc.sh > gpurun_out/${tag}_mfma.log 2>&1; tail -3 gpurun_out/${tag}_mfma.log
