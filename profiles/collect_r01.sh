#!/bin/bash
# Round-1 profile collection (run on the MI355X box through gpurun from the repo root; outputs under gpurun_out/$PROFILE_TAG, default p6).
# Kernel times and PMC counters are taken in SEPARATE runs; FETCH_SIZE and WRITE_SIZE need separate passes (TCC slots).
set -e
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/${PROFILE_TAG:-p6}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for m in fp32 f16x3; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$m -o s -- python3 $R/bench.py --math $m --steps 20 --warmup 5 --no-cpu-baseline --no-split > $O/stats_$m.log 2>&1
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch_$m -o p -- python3 $R/bench.py --math $m --steps 8 --warmup 2 --no-cpu-baseline --no-split > $O/fetch_$m.log 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write_$m -o p -- python3 $R/bench.py --math $m --steps 8 --warmup 2 --no-cpu-baseline --no-split > $O/write_$m.log 2>&1
done
find $O -name "*.csv" | head -20
