for i in 1 2; do
for v in 1 0; do
MP_GEMM_EP8=$v python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('ep8=$v', d['ms_per_step'], d['roofline']['frac'], d['roofline']['avg_launch_us'])"
done; done
