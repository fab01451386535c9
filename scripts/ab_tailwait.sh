# same-box A/B: the caller's stream ordered behind the mask tail's forward at the end of model_forward (MP_TAIL_WAIT=1, the round-1 behaviour) or only
# where a loss value is read (default)
for i in 1 2; do for v in 1 0; do
MP_TAIL_WAIT=$v python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('MP_TAIL_WAIT=$v', d['ms_per_step'], d['roofline']['frac'], d['loss_last'])"
done; done
