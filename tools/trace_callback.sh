#!/bin/bash
# Kernel trace of one callback loop (tools/pmc_callback.py): usage tools/trace_callback.sh <outdir> <what> [P] [n]
set -e
out=$1; what=$2; P=${3:-128}; n=${4:-20}
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$out/trace -o t -- python3 $R/tools/pmc_callback.py $what $P $n > $R/$out/trace.log 2>&1
cd $R
python3 tools/prof_summary.py $(find $out/trace -name '*kernel_trace.csv' | head -1) > $out/kernel_summary.txt
cut -c1-150 $out/kernel_summary.txt
