#!/bin/bash
# A/B of FormFunction with and without its gather pass (option gather_pass), with the bytes each kernel moved.
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for v in 0 1; do
  python3 $R/tools/elliptic_bench.py 256 gather_pass=$v 2>&1 | grep FormFunction
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout -k 10 200 rocprofv3 --pmc $c --output-format csv -d $R/gpurun_out/r3_ab/$v$c -o t -- python3 $R/tools/pmc_callback.py ell_fn 256 3 gather_pass=$v > /dev/null 2>&1
    python3 $R/tools/pmc_summary.py $(find $R/gpurun_out/r3_ab/$v$c -name '*counter_collection.csv' | head -1) | grep -E "fused4|gather|cprod" | cut -c1-140
  done
done
