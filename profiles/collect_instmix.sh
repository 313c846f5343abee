#!/bin/bash
# Instruction-mix PMC passes of one bench configuration (run through gpurun from the repo root):
#   profiles/collect_instmix.sh [bench args...]        e.g. profiles/collect_instmix.sh --config S256
# Round-4 finding (tools/microbench/mfma_valu_overlap.hip): v_mfma_f32_16x16x4_f32 does NOT overlap VALU instructions on its
# SIMD, so every VALU instruction of a layer loop is MFMA time lost -- these passes count them per kernel.
set +e
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/p_instmix
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
[ -f $O/counters.txt ] || rocprofv3 -L > $O/counters.txt 2>&1
B="$R/bench.py --no-cpu-baseline --no-split --no-other-configs --no-collective-probe --sustain-s 0 --steps 6 --warmup 2 --preheat-ms 0 $*"
timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES --output-format csv -d $O/a -o p -- python3 $B > $O/a.log 2>&1
timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE --output-format csv -d $O/b -o p -- python3 $B > $O/b.log 2>&1
cd $R
python3 - <<'PY'
import csv, collections, glob
for sub in ("a", "b"):
    f = glob.glob("gpurun_out/p_instmix/%s/*counter_collection.csv" % sub)
    if not f: print(sub, "no csv"); continue
    tot, cnt = collections.defaultdict(float), collections.defaultdict(int)
    for r in csv.DictReader(open(f[0])):
        k = (r["Kernel_Name"].split("(")[0][-44:], r["Counter_Name"])
        tot[k] += float(r["Counter_Value"]); cnt[k] += 1
    for k in sorted(tot):
        if any(s in k[0] for s in ("qnet_", "sage_", "head_", "csr_")):
            print(sub, k[0], k[1], round(tot[k] / cnt[k], 1), cnt[k])
PY
