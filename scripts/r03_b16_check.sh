#!/bin/bash
# B = 16 with the towers started ahead (the configuration that stalled ~1 s per step while two split-tail kernels could wait on two streams),
# against towers in order; then the default line.  Losses of the two B = 16 runs must agree.
set -u
mkdir -p gpurun_out
python -m pytest tests/test_gpu_trunk_kernels.py -q -m gpu -k "gemm320 or 320_row" -s 2>&1 | grep -E "concurrent|passed|failed|Error" | tail -5
for tag in ahead inorder; do
  extra=""; [ $tag = inorder ] && extra="--towers-in-order"
  timeout 300 python bench.py --batch 16 --steps 10 --warmup 5 --no-cpu-baseline --no-lora-line --roofline-steps 0 $extra > gpurun_out/b16_$tag.json 2> gpurun_out/b16_$tag.err
  python -c "
import json; d=json.load(open('gpurun_out/b16_$tag.json')); print('$tag', d['value'], d['ms_per_step'], d.get('loss_after_warmup'), d.get('loss_last'))"
done
timeout 300 python bench.py --batch 12 --steps 10 --warmup 5 --no-cpu-baseline --no-lora-line --roofline-steps 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('b12', d['value'], d['ms_per_step'])"
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r03f_bench.json 2> gpurun_out/r03f_bench.err
python -c "
import json; d=json.load(open('gpurun_out/r03f_bench.json')); print('default', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['all_bf16_gemms']['frac'], d['lora_stage3']['ms_per_step'])"
