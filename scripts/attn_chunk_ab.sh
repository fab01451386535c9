# workgroup order of the attention forward: (batch, head) chunks of 64 (default) vs the plain order (MP_ATTN_CHUNK=0) vs other chunk sizes
for c in 0 64 32 128 0 64; do echo -n "MP_ATTN_CHUNK=$c: "; MP_ATTN_CHUNK=$c python scripts/attn_pad_ab.py 2>/dev/null | grep "pad 320 fwd:" ; done
