#!/bin/bash
# rocprofv3 passes over the fused-upsampler micro-benchmark: kernel trace (durations) and PMC sets, each in its own run
export TMPDIR=/tmp
out=gpurun_out/up_pmc; mkdir -p $out
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python scripts/upsampler_bench.py > $out/trace.log 2>&1
i=0
for set in "SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS" "SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_MISC" "GRBM_GUI_ACTIVE SQ_WAIT_ANY SQ_INSTS_SALU SQ_INSTS_VMEM_WR"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $out/pmc$i -- python scripts/upsampler_bench.py > $out/pmc$i.log 2>&1
done
python - <<'PY'
import csv, glob, collections
for d in sorted(glob.glob('gpurun_out/up_pmc/*/')):
    for f in glob.glob(d + '**/*counter_collection.csv', recursive=True):
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            if 'upsample' in r['Kernel_Name']:
                agg[(r['Grid_Size'])][r['Counter_Name']].append(float(r['Counter_Value']))
        for g, cs in agg.items():
            print(d, 'grid', g, {k: round(sum(v) / len(v), 1) for k, v in cs.items()})
    for f in glob.glob(d + '**/*kernel_stats.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            if 'upsample' in r['Name']:
                print(d, r)
    for f in glob.glob(d + '**/*kernel_trace.csv', recursive=True):
        if 'trace/' in f:
            agg = collections.defaultdict(list)
            for r in csv.DictReader(open(f)):
                if 'upsample' in r['Kernel_Name']:
                    agg[r['Grid_Size']].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
            for g, v in agg.items():
                v.sort(); print('durations grid', g, 'n', len(v), 'median ns', v[len(v)//2], 'min', v[0])
PY
