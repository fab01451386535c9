#!/bin/bash
# round 4: the new bench line end to end (secondary objects, parity gate), the ep code path on one GPU, the touched GPU tests
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
./scripts/lab/ups_lab > gpurun_out/r04_ups_timeline.txt 2>&1; tail -45 gpurun_out/r04_ups_timeline.txt
t0=$(date +%s)
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r04b_bench.json 2> gpurun_out/r04b_bench.err
echo "== bench rc=$? wall $(( $(date +%s) - t0 )) s"; tail -3 gpurun_out/r04b_bench.err | cut -c1-400
python - <<'PY'
import json
d=json.load(open('gpurun_out/r04b_bench.json'))
print(d['ms_per_step'], d['value'], d['roofline']['frac'], d['roofline'].get('expert_rows'), d['roofline'].get('traffic_detail',{}).get('stale'))
print('configs', json.dumps(d.get('configs'))[:900]); print('decode', d.get('decode')); print('cpu', d.get('cpu_baseline'))
print('lora', d['lora_stage3'].get('ms_per_step'), 'ups', d['roofline_upsampler']['sam1024'], d['roofline_upsampler']['copy_floor'].get('upsampler_vs_copy'))
PY
python bench.py --gpus 1 --ep 1 --steps 6 --warmup 2 > gpurun_out/r04b_ep1.json 2> gpurun_out/r04b_ep1.err; echo "== ep1 rc=$?"; cut -c1-1500 gpurun_out/r04b_ep1.json; tail -2 gpurun_out/r04b_ep1.err | cut -c1-300
MP_BENCH_FORCE_DIST=1 python bench.py --gpus 1 --ep 1 --ep-comm capi --ep-variable --steps 4 --warmup 1 > gpurun_out/r04b_ep1v.json 2> gpurun_out/r04b_ep1v.err; echo "== ep1 capi variable rc=$?"; python -c "
import json; d=json.load(open('gpurun_out/r04b_ep1v.json')); print(d['ms_per_step'], d['ep'], d['rccl_ranks'])"; tail -2 gpurun_out/r04b_ep1v.err | cut -c1-300
python -m pytest tests/test_gpu_model.py -x -q -k "evaluate or full_depth or fused or golden" 2>&1 | tail -6
python -m pytest tests/test_gpu_mask_tail_kernels.py -x -q 2>&1 | tail -3
