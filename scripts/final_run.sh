cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
bash scripts/gpu_tests.sh
echo "== tests rc=$?"
python bench.py > gpurun_out/bench_l.json 2> gpurun_out/bench_l.err; tail -3 gpurun_out/bench_l.err; cat gpurun_out/bench_l.json | cut -c1-400
rm -rf gpurun_out/prof_l; timeout 900 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_l -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/bench_l_prof.json 2> gpurun_out/bench_l_prof.err
ls gpurun_out/prof_l/*/ | head
db=$(ls gpurun_out/prof_l/*/*.db | head -1); python scripts/rocpd_stats.py $db 4 gpurun_out/r01l_kernel_stats.md; head -30 gpurun_out/r01l_kernel_stats.md
# keep the db out of the merge budget
rm -rf gpurun_out/prof_l
