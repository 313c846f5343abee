#!/bin/bash
# PMC view of the layer-major kernels on the ragged MIX batch (run through gpurun from the repo root).
set +e
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/p_mix
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
# (no RCCL probe and no 2-s window under the profiler; every pass bounded: a pass of this script once hung on a box)
B="$R/bench.py --config MIX --no-cpu-baseline --no-split --no-collective-probe --sustain-s 0 --eager --steps 6 --warmup 2 --preheat-ms 0"
timeout -k 10 150 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch -o p -- python3 $B > $O/fetch.log 2>&1
timeout -k 10 150 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write -o p -- python3 $B > $O/write.log 2>&1
timeout -k 10 150 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/l2 -o p -- python3 $B > $O/l2.log 2>&1
timeout -k 10 150 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES --output-format csv -d $O/mfma -o p -- python3 $B > $O/mfma.log 2>&1
timeout -k 10 150 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS --output-format csv -d $O/sq -o p -- python3 $B > $O/sq.log 2>&1
cd $R
python3 - <<'PY'
import csv, collections, glob
for sub in ("fetch","write","l2","mfma","sq"):
    f = glob.glob("gpurun_out/p_mix/%s/*counter_collection.csv" % sub)
    if not f: print(sub, "no csv"); continue
    tot, cnt = collections.defaultdict(float), collections.defaultdict(int)
    for r in csv.DictReader(open(f[0])):
        k = (r["Kernel_Name"].split("(")[0][-40:], r["Counter_Name"])
        tot[k] += float(r["Counter_Value"]); cnt[k] += 1
    for k in sorted(tot):
        if any(s in k[0] for s in ("sage_hidden", "sage_stack", "head_fwd", "head_bwd", "sage_dw_kernel")):
            print(sub, k[0], k[1], round(tot[k] / cnt[k], 1), cnt[k])
PY
