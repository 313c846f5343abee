#!/bin/bash
# Round-3 profile collection (run on the MI355X box through gpurun from the repo root; outputs under
# gpurun_out/$PROFILE_TAG, default p_r03).  Kernel times and PMC counters are taken in SEPARATE runs; FETCH_SIZE and
# WRITE_SIZE need separate passes (TCC slots); the SQ passes (MFMA busy cycles, LDS bank conflicts) are their own runs too.
# The program after `--` is python3 itself (no env / shell hop under the profiler).
set -e
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/${PROFILE_TAG:-p_r03}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $O/counters_available.txt 2>&1 || true
B="$R/bench.py --no-cpu-baseline --no-split --no-other-configs --no-collective-probe"
for m in ${PROFILE_MODES:-fp32}; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$m -o s -- python3 $B --math $m --steps 20 --warmup 5 > $O/stats_$m.log 2>&1
  echo "stats $m done"
  [ -n "$PROFILE_SKIP_PMC" ] && continue        # (re-collection of the kernel times only: the PMC figures stay valid)
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch_$m -o p -- python3 $B --math $m --steps 8 --warmup 2 > $O/fetch_$m.log 2>&1
  echo "fetch $m done"
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write_$m -o p -- python3 $B --math $m --steps 8 --warmup 2 > $O/write_$m.log 2>&1
  echo "write $m done"
  rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/mfma_$m -o p -- python3 $B --math $m --steps 8 --warmup 2 > $O/mfma_$m.log 2>&1
  echo "mfma $m done"
  rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY --output-format csv -d $O/lds_$m -o p -- python3 $B --math $m --steps 8 --warmup 2 > $O/lds_$m.log 2>&1
  echo "lds $m done"
done
find $O -name "*.csv" | head -40
# round 3 extras: the other single-GPU BASELINE configurations (kernel stats) and the bench lines themselves
cd /tmp
for c in S256 MIX; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$c -o s -- python3 $B --config $c --steps 20 --warmup 5 > $O/stats_$c.log 2>&1
  echo "stats $c done"
done
cd $R
python3 bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err
python3 bench.py --steps 20 --warmup 5 --eager --no-cpu-baseline --no-other-configs --no-split > $O/bench_eager.json 2>/dev/null
python3 bench.py --steps 20 --warmup 5 --config S256 --eager --no-cpu-baseline --no-split > $O/bench_S256_eager.json 2>/dev/null
python3 bench.py --steps 20 --warmup 5 --config S256 --no-cpu-baseline --no-split > $O/bench_S256.json 2>/dev/null
python3 bench.py --steps 20 --warmup 5 --config MIX --no-cpu-baseline --no-split > $O/bench_MIX.json 2>/dev/null
python3 bench.py --steps 20 --warmup 5 --data D1 --no-cpu-baseline --no-split --no-other-configs > $O/bench_D1.json 2>/dev/null
echo "bench lines done"
# only summaries travel back (gpurun merges at most 64 MiB): kernel-stats CSVs, the PMC figures as JSON, the bench lines
S=$R/gpurun_out/r03_summary
mkdir -p $S
cp $O/stats_fp32/s_kernel_stats.csv $S/kernel_stats_fp32.csv
cp $O/stats_S256/s_kernel_stats.csv $S/kernel_stats_S256.csv
cp $O/stats_MIX/s_kernel_stats.csv $S/kernel_stats_MIX.csv
[ -z "$PROFILE_SKIP_PMC" ] && python3 profiles/pmc_to_json.py $O fp32 > $S/traffic_pmc.json
cp $O/bench_*.json $S/
rm -rf $O
ls -la $S
