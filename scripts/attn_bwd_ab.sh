for st in 2 4 2 4; do echo "MP_ATTN_BWD_STAGES=$st"; MP_ATTN_BWD_STAGES=$st python scripts/attn_bwd_bench.py 2>/dev/null | grep -v "^$"; done
