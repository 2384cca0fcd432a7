#!/bin/bash
# Round-6 record run (on the GPU box): full -m gpu suite, the round profile of the headline (kernel trace, counter passes, traffic record,
# default bench line), the 64^3 Stokes callback counters, one slab rank's kernel trace at G = 8 / 2, the stamp record.
set -e
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
export GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$PWD}
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/r06_gputest.log 2>&1 || { tail -40 gpurun_out/r06_gputest.log; exit 1; }
tail -3 gpurun_out/r06_gputest.log
bash tools/profile_round.sh r06 > gpurun_out/r06_profile_round.log 2>&1 || { tail -30 gpurun_out/r06_profile_round.log; exit 1; }
echo "profile round done"
bash tools/pmc_callback_passes.sh gpurun_out/r06_pmc_callbacks/stokes_lin64 stokes_lin 64 8 > gpurun_out/r06_pmc_callbacks_64.log 2>&1 || { tail -20 gpurun_out/r06_pmc_callbacks_64.log; exit 1; }
echo "callback counters done"
bash tools/r06_t9.sh > gpurun_out/r06_rank_trace.log 2>&1 || { tail -20 gpurun_out/r06_rank_trace.log; exit 1; }
( timeout -k 10 120 python tools/stamp_probe3.py 256 tools/libchebhip_diag.so; timeout -k 10 120 python tools/stamp_probe3.py 256 tools/libchebhip_diag_ov0.so ) 2>&1 | grep -v amdgpu.ids > gpurun_out/r06_stamps_final.txt
python3 bench.py --steps 20 --warmup 5 > gpurun_out/r06_bench_driver_style.json 2> gpurun_out/r06_bench_driver_style.err
tail -c 600 gpurun_out/r06_bench.json
