#!/bin/bash
# kernel-time summary of the MIX step (run through gpurun from the repo root): tools/prof_mix.sh [tag]
R=${GRAFT_REPO_ROOT:-$PWD}
T=${1:-mix}
O=$R/gpurun_out/prof_$T
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o p -- python3 $R/bench.py --config MIX --no-cpu-baseline --no-split --no-collective-probe --sustain-s 0 --steps 20 --warmup 5 --preheat-ms 0 > $O/run.log 2>&1
cd $R
python3 - "$O" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)
if not f: print("no stats"); sys.exit(0)
rows = list(csv.DictReader(open(f[0])))
for r in rows[:12]:
    print("%-44s calls %6s avg %9.1f us  total %9.1f ms" % (r["Name"].split("(")[0][-44:], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
PY
find $O -name "*.csv" ! -name "*kernel_stats.csv" -delete
