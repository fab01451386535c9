#!/bin/bash
# PMC passes over the GEMM micro-benchmark (each counter set in its own rocprofv3 run; kernel-trace only, no sys-trace)
export TMPDIR=/tmp
mkdir -p gpurun_out/pmc
for ver in 3; do
for set in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS GRBM_GUI_ACTIVE"; do
  tag=v${ver}_$(echo $set | tr ' ' '_' | cut -c1-30)
  MP_GEMM_VARIANT=2 timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d gpurun_out/pmc/$tag -- python scripts/gemm_bench.py --child > gpurun_out/pmc/$tag.log 2>&1
done; done
