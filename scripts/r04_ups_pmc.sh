#!/bin/bash
# round 4: PMC counters of the shipped fused upsampler at the 1024-px geometry (batch 8), one rocprofv3 pass per counter set (kernel-trace only)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; out=gpurun_out/r04_ups_pmc; rm -rf $out; mkdir -p $out
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_ADDR_CONFLICT SQ_INSTS_SALU" \
           "GRBM_GUI_ACTIVE SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_MISC SQ_LDS_UNALIGNED_STALL SQ_LDS_MEM_VIOLATIONS"; do
  i=$((i+1))
  timeout 200 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $out/p$i -- ./scripts/lab/ups_lab 64 pmc > $out/p$i.log 2>&1 || tail -3 $out/p$i.log
done
python - <<'PY'
import csv, glob, collections, json
res = {}
for f in sorted(glob.glob('gpurun_out/r04_ups_pmc/**/*counter_collection.csv', recursive=True)):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if 'upsample_fused_kernel' in r['Kernel_Name']:
            agg[r['Counter_Name']].append(float(r['Counter_Value']))
    for k, v in agg.items():
        res[k] = round(sum(v[5:]) / max(len(v[5:]), 1), 1)
print(json.dumps(res, indent=1))
json.dump(res, open('gpurun_out/r04_ups_pmc.json', 'w'), indent=1)
PY
