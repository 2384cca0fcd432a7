set -e
mkdir -p gpurun_out/r06_stokes_trace
export PYTHONUNBUFFERED=1
R=$PWD
cd /tmp && export TMPDIR=/tmp
for V in 0 4; do
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r06_stokes_trace/pk$V -o t -- python3 $R/tools/stokes_rank_trace.py 2 40 dist_packed_exchange=$V > $R/gpurun_out/r06_stokes_trace/pk$V.log 2>&1
done
cd $R
for V in 0 4; do
python3 tools/prof_summary.py $(find gpurun_out/r06_stokes_trace/pk$V -name '*kernel_trace.csv' | head -1) --skip 21 > gpurun_out/r06_stokes_trace/kernels_pk$V.txt
echo "== dist_packed_exchange=$V"; cut -c1-170 gpurun_out/r06_stokes_trace/kernels_pk$V.txt | head -14
done
