# same-box A/B: the frozen towers of a step started beside the previous step's decoder (bench.py --towers-ahead = model.towers_run_ahead) or queued behind it
for i in 1 2; do for cfg in "" "--towers-ahead"; do
python bench.py --steps 20 --warmup 5 --no-cpu-baseline $cfg 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('[$cfg]', d['ms_per_step'], d['value'], d['roofline']['frac'], d['roofline']['avg_launch_us'])"
done; done
