#!/bin/bash
# Counter pass of tools/lds_probe (one kernel per LDS access pattern of the 256-point sweep kernel): conflict cycles per pattern.
# usage (on the GPU box): bash tools/r06_lds_probe.sh <outdir>
set -e
out=${1:-gpurun_out/r06_lds_probe}
mkdir -p $out
R=$PWD
cd /tmp && export TMPDIR=/tmp
timeout -k 10 240 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL --output-format csv -d $R/$out/pmc -o t -- $R/tools/lds_probe > $R/$out/pmc.log 2>&1 || { tail -20 $R/$out/pmc.log; exit 1; }
cd $R
f=$(find $out/pmc -name '*counter_collection.csv' | head -1)
python3 tools/pmc_summary.py $f > $out/summary.txt
cat $out/summary.txt
