#!/bin/bash
# Round-4 profile collection (run on the MI355X box through gpurun from the repo root): for EVERY single-GPU BASELINE
# configuration (L256 start positions, S256, MIX) the kernel-trace statistics and the PMC passes -- FETCH_SIZE and WRITE_SIZE in
# separate runs (TCC slots), MFMA busy cycles and the LDS / wait counters in their own runs; never --pmc together with a
# trace domain other than --kernel-trace.  The program after `--` is python3 itself.  Every pass is bounded.
set +e
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/p_r04
S=$R/gpurun_out/r04_summary
mkdir -p $O $S
cd /tmp && export TMPDIR=/tmp
for c in ${PROFILE_CONFIGS:-L256 S256 MIX}; do
  B="$R/bench.py --config $c --no-cpu-baseline --no-split --no-other-configs --no-collective-probe --sustain-s 0"
  timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$c -o s -- python3 $B --steps 20 --warmup 5 > $O/stats_$c.log 2>&1
  cp $O/stats_$c/s_kernel_stats.csv $S/kernel_stats_$c.csv
  echo "stats $c done"
  [ -n "$PROFILE_SKIP_PMC" ] && continue
  P="$B --steps 8 --warmup 2 --preheat-ms 0"
  timeout -k 10 240 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch_$c -o p -- python3 $P > $O/fetch_$c.log 2>&1
  timeout -k 10 240 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write_$c -o p -- python3 $P > $O/write_$c.log 2>&1
  timeout -k 10 240 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/mfma_$c -o p -- python3 $P > $O/mfma_$c.log 2>&1
  timeout -k 10 240 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY --output-format csv -d $O/lds_$c -o p -- python3 $P > $O/lds_$c.log 2>&1
  echo "pmc $c done"
done
cd $R
[ -z "$PROFILE_SKIP_PMC" ] && python3 profiles/pmc_to_json.py --by-tag $O ${PROFILE_CONFIGS:-L256 S256 MIX} > $S/traffic_pmc.json
python3 bench.py --steps 20 --warmup 5 > $S/bench_default.json 2> $S/bench_default.err
python3 bench.py --steps 20 --warmup 5 --config S256 --no-cpu-baseline --no-split > $S/bench_S256.json 2>/dev/null
python3 bench.py --steps 20 --warmup 5 --config MIX --no-cpu-baseline --no-split > $S/bench_MIX.json 2>/dev/null
python3 bench.py --steps 20 --warmup 5 --data D1 --no-cpu-baseline --no-split --no-other-configs > $S/bench_D1.json 2>/dev/null
python3 bench.py --steps 20 --warmup 5 --eager --no-cpu-baseline --no-other-configs --no-split > $S/bench_eager.json 2>/dev/null
python3 bench.py --steps 20 --warmup 5 --config S256 --eager --no-cpu-baseline --no-split > $S/bench_S256_eager.json 2>/dev/null
python3 bench.py --steps 20 --warmup 5 --config S256 --eager --plain-autograd --no-cpu-baseline --no-split > $S/bench_S256_eager_plain.json 2>/dev/null
echo "bench lines done"
rm -rf $O/*/*.db 2>/dev/null
ls -la $S
