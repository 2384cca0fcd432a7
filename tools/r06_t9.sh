set -e
mkdir -p gpurun_out/r06_rank_trace
export PYTHONUNBUFFERED=1
R=$PWD
cd /tmp && export TMPDIR=/tmp
for G in 8 2; do
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r06_rank_trace/G$G -o t -- python3 $R/tools/dist_rank_trace.py $G 50 > $R/gpurun_out/r06_rank_trace/G$G.log 2>&1
done
cd $R
for G in 8 2; do
python3 tools/prof_summary.py $(find gpurun_out/r06_rank_trace/G$G -name '*kernel_trace.csv' | head -1) > gpurun_out/r06_rank_trace/rank_kernels_G$G.txt
cut -c1-150 gpurun_out/r06_rank_trace/rank_kernels_G$G.txt | head -12
done
