# same-box A/B of LoRA-step switches: MP_LORA_SWIGLU_KEEP, MP_TN_SKINNY_MFMA, MP_GEMM320
for i in 1 2; do
for cfg in "" "MP_LORA_SWIGLU_KEEP=0" "MP_GEMM320=0"; do
echo -n "[$cfg] "; env $cfg python bench.py --lora --steps 10 --warmup 3 2>&1 | grep "gpu leg"
done; done
