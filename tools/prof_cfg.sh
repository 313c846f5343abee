#!/bin/bash
# kernel-time summary of one bench configuration (run through gpurun from the repo root): tools/prof_cfg.sh CONFIG [tag]
R=${GRAFT_REPO_ROOT:-$PWD}
C=${1:-L256}
T=${2:-$C}
O=$R/gpurun_out/prof_$T
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o p -- python3 $R/bench.py --config $C --no-cpu-baseline --no-split --no-other-configs --no-collective-probe --sustain-s 0 --steps 20 --warmup 5 --preheat-ms 0 > $O/run.log 2>&1
cd $R
python3 - "$O" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)
if not f: print("no stats"); sys.exit(0)
for r in list(csv.DictReader(open(f[0])))[:10]:
    print("%-44s calls %6s avg %9.1f us" % (r["Name"].split("(")[0][-44:], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
find $O -name "*.csv" ! -name "*kernel_stats.csv" -delete
