#!/bin/bash
# Kernel trace + counter passes (one rocprofv3 run per counter set; FETCH_SIZE and WRITE_SIZE alone) of one operator
# callback loop (tools/pmc_callback.py).  usage: tools/pmc_callback_passes.sh <outdir> <what> [P] [n]
set -e
out=$1; what=$2; P=${3:-128}; n=${4:-6}
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$out/trace -o t -- python3 $R/tools/pmc_callback.py $what $P 20 > $R/$out/trace.log 2>&1
run() { name=$1; shift; timeout -k 10 240 rocprofv3 --pmc "$@" --output-format csv -d $R/$out/$name -o t -- python3 $R/tools/pmc_callback.py $what $P $n > $R/$out/$name.log 2>&1; }
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU
run fetch FETCH_SIZE
run write WRITE_SIZE
cd $R
python3 tools/prof_summary.py $(find $out/trace -name '*kernel_trace.csv' | head -1) > $out/kernel_summary.txt
for nme in sq1 fetch write; do f=$(find $out/$nme -name '*counter_collection.csv' | head -1); [ -n "$f" ] && python3 tools/pmc_summary.py $f ; done > $out/counters.txt
cat $out/kernel_summary.txt
