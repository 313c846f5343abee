"""Turn the rocprofv3 --pmc passes of profiles/collect_r0N.sh into per-launch figures for the three big kernels.

* FETCH_SIZE / WRITE_SIZE (KiB; on gfx950 FETCH_SIZE is doubled: MI355X_MICROARCH.md, HBM / rocprofv3 section) -> HBM bytes
* SQ_VALU_MFMA_BUSY_CYCLES against GRBM_GUI_ACTIVE (sum over the 8 XCDs -> / 8 = shader cycles of the dispatch) and the
  chip's 1024 SIMDs -> fraction of the matrix pipe's cycles that were busy
* SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE -> share of LDS-array cycles lost to bank conflicts; SQ_WAIT_ANY / SQ_WAVE_CYCLES

Usage: python profiles/pmc_to_json.py gpurun_out/p_r02 fp32 f16x3 > profiles/r02/traffic_pmc.json
       python profiles/pmc_to_json.py --by-tag gpurun_out/p_r04 L256 S256 MIX > profiles/r04/traffic_pmc.json
       (round 4: one entry per bench configuration, {"L256": {kernel: ...}, "S256": ..., "MIX": ...}; bench.py reads the entry of
       the configuration it runs)
"""
import collections
import csv
import json
import os
import sys

KERNELS = ("qnet_fwd_kernel", "qnet_bwd_kernel", "sage_dw_kernel", "sage_dw16_kernel", "sage_hidden_fwd_kernel",
           "sage_hidden_bwd_kernel", "sage_stack_fwd_kernel", "sage_stack_bwd_kernel")
N_SIMD = 256 * 4


def per_kernel(path, counter):
    tot, cnt = collections.defaultdict(float), collections.defaultdict(int)
    if not os.path.exists(path):
        return {}
    with open(path) as f:
        for row in csv.DictReader(f):
            if row["Counter_Name"] != counter:
                continue
            name = row["Kernel_Name"]
            tot[name] += float(row["Counter_Value"])
            cnt[name] += 1
    return {k: tot[k] / cnt[k] for k in tot}


def main(root, mode):
    def load(sub, counter):
        return per_kernel("%s/%s_%s/p_counter_collection.csv" % (root, sub, mode), counter)
    fetch, write = load("fetch", "FETCH_SIZE"), load("write", "WRITE_SIZE")
    busy, gui = load("mfma", "SQ_VALU_MFMA_BUSY_CYCLES"), load("mfma", "GRBM_GUI_ACTIVE")
    conf, idx = load("lds", "SQ_LDS_BANK_CONFLICT"), load("lds", "SQ_LDS_IDX_ACTIVE")
    wcyc, wany = load("lds", "SQ_WAVE_CYCLES"), load("lds", "SQ_WAIT_ANY")
    out = {}
    for name in fetch:
        if not any(k in name for k in KERNELS):
            continue
        short = name.split("(")[0]
        fr = fetch[name] * 1024.0
        wr = write.get(name, 0.0) * 1024.0
        ent = {"fetch_bytes_raw": fr, "fetch_bytes_corrected": 2.0 * fr, "write_bytes": wr,
               "hbm_bytes_per_launch": 2.0 * fr + wr,
               "note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, KiB units); FETCH_SIZE doubled for "
                       "gfx950 (MI355X_MICROARCH.md, HBM)"}
        if name in busy and gui.get(name):
            cycles = gui[name] / 8.0
            ent["mfma_busy_cycles"] = busy[name]
            ent["dispatch_cycles"] = cycles
            ent["mfma_pipe_busy_frac"] = busy[name] / (N_SIMD * cycles)
        if name in conf and idx.get(name):
            ent["lds_bank_conflict_frac"] = conf[name] / idx[name]
        if name in wany and wcyc.get(name):
            ent["wave_wait_frac"] = wany[name] / wcyc[name]
        out[short] = ent
    return out


if __name__ == "__main__":
    if sys.argv[1] == "--by-tag":
        root = sys.argv[2]
        print(json.dumps({tag: main(root, tag) for tag in sys.argv[3:]}, indent=1))
        sys.exit(0)
    root = sys.argv[1]
    res = {}
    for mode in sys.argv[2:]:
        res.update(main(root, mode))
    print(json.dumps(res, indent=1))
