#!/bin/bash
# round 4: every -m gpu module, smoke, the driver's bench command
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
tag=${1:-r04c}
python -m pytest tests -m gpu -x -q 2>&1 | tail -8
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
t0=$(date +%s)
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err
echo "== bench rc=$? wall $(( $(date +%s) - t0 )) s"; grep "gpu leg\|PARITY" gpurun_out/${tag}_bench.err
python - <<PY
import json
d=json.load(open('gpurun_out/${tag}_bench.json'))
print(d['ms_per_step'], d['value'], d['roofline']['frac'], d['roofline'].get('all_bf16_gemms',{}).get('frac'))
print('lora', d['lora_stage3'].get('ms_per_step'), 'ups', d['roofline_upsampler']['sam1024'], d['roofline_upsampler']['sam256'], d['roofline_upsampler']['copy_floor'].get('upsampler_vs_copy'))
print('parity', d['parity']['mask']['max_abs_dlogit'], d['parity']['abs_dloss'], d['parity']['hidden_rel_err'])
PY
python scripts/decode_bench.py 2>&1 | tail -1
