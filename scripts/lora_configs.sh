# the shipped scripts' LoRA configurations at true dimensions (scripts/lora_bench.py): stage III (dense, gate/up/down r 8), stage II (all seven targets,
# r 16, + the whole --sft_modules set), stage IV (MoE E = 2: per-expert adapters, q / v adapters, wg, lm_head, embed_tokens)
python scripts/lora_bench.py 2>/dev/null | tail -1
python scripts/lora_bench.py --lora_r 16 --targets q_proj,k_proj,v_proj,o_proj,gate_proj,up_proj,down_proj 2>/dev/null | tail -1
python scripts/lora_bench.py --lora_r 16 --targets q_proj,k_proj,v_proj,o_proj,gate_proj,up_proj,down_proj --sft mask_decoder,text_hidden_fcs,lm_head,embed_tokens,input_layernorm,post_attention_layernorm,mm_projector 2>/dev/null | tail -1
python scripts/lora_bench.py --moe --targets gate_proj,up_proj,down_proj,q_proj,v_proj --sft mask_decoder,text_hidden_fcs,lm_head,embed_tokens,wg 2>/dev/null | tail -1
