#!/bin/bash
# Counter passes of the Poisson matvec, one rocprofv3 run per counter set (SQ sets of <= 8, FETCH_SIZE and
# WRITE_SIZE alone: MI355X_MICROARCH.md "rocprofv3 PMC slots").  The TCC/TCP/TA stall sets that hung rocprofv3 in
# round 1 are deliberately absent.  usage: tools/pmc_passes.sh <outdir> [P]
set -e
out=$1; P=${2:-256}
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
run() { name=$1; shift; timeout -k 10 240 rocprofv3 --pmc "$@" --output-format csv -d $R/$out/$name -o t -- python3 $R/tools/pmc_matvec.py $P 12 > $R/$out/$name.log 2>&1; }
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU
run sq2 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_VALU_MFMA_COEXEC_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
run fetch FETCH_SIZE
run write WRITE_SIZE
run grbm GRBM_GUI_ACTIVE
cd $R
for n in sq1 sq2 fetch write grbm; do f=$(find $out/$n -name '*counter_collection.csv' | head -1); [ -n "$f" ] && python3 tools/pmc_summary.py $f | grep cheb_ ; done > $out/summary.txt
cat $out/summary.txt
