# same-box A/B of the 320-row tile kernel: the decoder's shapes alone (sustained), then the whole training step
for v in 0 1; do echo "MP_GEMM320=$v"; MP_GEMM320=$v python scripts/gemm_sustained.py 4 4 1 2>&1 | tail -1; done
for i in 1 2; do
for v in 0 1; do
MP_GEMM320=$v python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('MP_GEMM320=$v', d['ms_per_step'], d['roofline']['frac'], d['roofline']['avg_launch_us'], d['roofline']['all_bf16_gemms']['frac'])"
done; done
