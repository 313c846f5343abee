#!/bin/bash
# Round-2 profile collection (run on the MI355X box through gpurun from the repo root; outputs under
# gpurun_out/$PROFILE_TAG, default p_r02).  Kernel times and PMC counters are taken in SEPARATE runs; FETCH_SIZE and
# WRITE_SIZE need separate passes (TCC slots); the SQ passes (MFMA busy cycles, LDS bank conflicts) are their own runs too.
# The program after `--` is python3 itself (no env / shell hop under the profiler).
set -e
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/${PROFILE_TAG:-p_r02}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $O/counters_available.txt 2>&1 || true
B="$R/bench.py --no-cpu-baseline --no-split --no-other-configs"
for m in ${PROFILE_MODES:-fp32 f16x3}; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$m -o s -- python3 $B --math $m --steps 20 --warmup 5 > $O/stats_$m.log 2>&1
  echo "stats $m done"
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
