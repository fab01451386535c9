for i in 1 2; do for v in 0 1; do
MP_BENCH_HIPRIO=$v python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timer 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('HIPRIO=$v', d['ms_per_step'], d['loss_last'])"
done; done
