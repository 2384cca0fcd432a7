#!/bin/bash
# Round profile of the headline bench: kernel trace + stats, then the counter passes (separate runs), then the
# traffic record bench.py reads (profiles/traffic.json, tied to the kernel sources by their hash).
# usage (on the GPU box): tools/profile_round.sh r02      -> writes gpurun_out/<tag>_*; copy what is to be judged into profiles/
set -e
tag=${1:-r03}
R=$GRAFT_REPO_ROOT; out=gpurun_out
cd /tmp && export TMPDIR=/tmp
# the traced run is the metric alone (no CPU baseline, no extras): spin-up 100 + warm-up 20 are dropped from the summary,
# which then covers exactly the 200 timed steps (3 launches each) that ms_per_step of the same run is taken over
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$out/${tag}_trace -o t -- python3 $R/bench.py --no-cpu-baseline --no-extras > $R/$out/${tag}_bench_under_rocprof.json 2> $R/$out/${tag}_trace.log
cd $R
f=$(find $out/${tag}_trace -name '*kernel_trace.csv' | head -1)
python3 tools/prof_summary.py $f --skip 120 > $out/${tag}_bench_kernel_summary.txt
tools/pmc_passes.sh $out/${tag}_pmc 256 > $out/${tag}_pmc.log 2>&1
python3 - <<PY
import json, re, sys
sys.path.insert(0, "$R")
import bench
tot = {}
for line in open("$R/$out/${tag}_pmc/summary.txt"):
    m = re.search(r"cheb_sweep_vec4_kernel<32, (\w+), (\w+)(?:, \d+)*>.*?(FETCH_SIZE|WRITE_SIZE)\s+n=\s*\d+ mean=\s*([\d.]+)", line)   # <KS, JFAST, MODE (0 store, 1 acc), RAW, INM>
    if m:
        k = (m.group(1), m.group(2)); tot.setdefault(k, 0.0)
        tot[k] += float(m.group(4)) * 1024.0 * (2.0 if m.group(3) == "FETCH_SIZE" else 1.0)   # KiB; FETCH_SIZE x2 on gfx950
per = [tot.get(("false", "0"), 0), tot.get(("false", "1"), 0), tot.get(("true", "1"), 0)]
assert all(v > 0 for v in per), "no FETCH_SIZE / WRITE_SIZE rows for the three launches of the matvec: %r" % (tot,)
rec = {"P": 256, "kernel": "cheb_sweep_vec4_kernel", "launches_per_matvec": 3, "hbm_bytes_per_matvec": sum(per), "hbm_bytes_per_launch": sum(per) / 3.0,
       "per_direction_bytes": per, "csrc_sha256": bench.csrc_hash(),
       "source": "rocprofv3 --pmc FETCH_SIZE, WRITE_SIZE in separate passes (tools/pmc_passes.sh); FETCH_SIZE x2 gfx950 correction; KiB units"}
json.dump(rec, open("$R/$out/${tag}_traffic.json", "w"), indent=1)
print(rec)
PY
python3 bench.py > $out/${tag}_bench.json 2> $out/${tag}_bench.err
cat $out/${tag}_bench.json
