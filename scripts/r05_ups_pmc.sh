#!/bin/bash
# round 5: the front-end counters the round-4 review asked for on the shipped fused upsampler (1024-px geometry, batch 8): instruction-cache
# requests / hits / misses, instruction fetches and their occupancy, issue stalls — one rocprofv3 pass per counter set (kernel-trace only).
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; out=gpurun_out/r05_ups_pmc; rm -rf $out; mkdir -p $out
hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off scripts/lab/ups_lab.hip medplib_amd/csrc/capi.cpp -o scripts/lab/ups_lab 2> $out/build.log || tail -5 $out/build.log
i=0
for set in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQC_ICACHE_BUSY_CYCLES SQC_TC_INST_REQ SQC_TC_STALL" \
           "SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU" \
           "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INST_LEVEL_SMEM SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 200 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $out/p$i -- ./scripts/lab/ups_lab 64 pmc > $out/p$i.log 2>&1 || tail -3 $out/p$i.log
done
python - <<'PY'
import csv, glob, collections, json
res = {}
for f in sorted(glob.glob('gpurun_out/r05_ups_pmc/**/*counter_collection.csv', recursive=True)):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if 'upsample_fused_kernel' in r['Kernel_Name']:
            agg[r['Counter_Name']].append(float(r['Counter_Value']))
    for k, v in agg.items():
        res[k] = round(sum(v[5:]) / max(len(v[5:]), 1), 1)
print(json.dumps(res, indent=1))
json.dump(res, open('gpurun_out/r05_upsampler_frontend_pmc.json', 'w'), indent=1)
PY
