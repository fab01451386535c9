for v in 0 1; do echo "MP_GEMM320=$v"; MP_GEMM320=$v python scripts/tower_bench.py 2>&1 | python -c "
import sys,json
t=sys.stdin.read(); d=json.loads(t[t.index('{'):])
print(d['clip_tower_plus_projector_ms'], d['clip_gemm_ms_total'])
for k,v in d['clip_gemms'].items(): print('   ',k,v)"; done
