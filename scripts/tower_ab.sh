for cfg in "" "MP_GEMM256_MIN_TILES=64" "MP_GEMM_SHORTK_RULE=0" "MP_GEMM256_MIN_TILES=64 MP_GEMM_SHORTK_RULE=0"; do
echo "== $cfg"
env $cfg python scripts/tower_bench.py 2>&1 | python -c "
import sys,json
t=sys.stdin.read(); d=json.loads(t[t.index('{'):])
print(d['clip_tower_plus_projector_ms'], d['sam_encoder_ms'], d['clip_gemm_ms_total'])
for k,v in d['clip_gemms'].items(): print('   ',k,v)"
done
