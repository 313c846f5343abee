"""Turn the rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of profiles/collect_r01.sh into per-launch HBM bytes.

FETCH_SIZE / WRITE_SIZE are reported in KiB; on gfx950 FETCH_SIZE is doubled (MI355X_MICROARCH.md, HBM / rocprofv3
section).  Usage: python profiles/pmc_to_json.py gpurun_out/p5 fp32 > profiles/r01/traffic_pmc.json
"""
import csv, json, sys, collections

def per_kernel(path, counter):
    tot, cnt = collections.defaultdict(float), collections.defaultdict(int)
    with open(path) as f:
        for row in csv.DictReader(f):
            if row["Counter_Name"] != counter:
                continue
            name = row["Kernel_Name"]
            tot[name] += float(row["Counter_Value"])
            cnt[name] += 1
    return {k: tot[k] / cnt[k] for k in tot}

def main(root, mode):
    fetch = per_kernel("%s/fetch_%s/p_counter_collection.csv" % (root, mode), "FETCH_SIZE")
    write = per_kernel("%s/write_%s/p_counter_collection.csv" % (root, mode), "WRITE_SIZE")
    out = {}
    for name in fetch:
        if not any(k in name for k in ("qnet_fwd_kernel", "qnet_bwd_kernel", "sage_dw_kernel", "sage_dw16_kernel")):
            continue
        short = name.split("(")[0]
        fr = fetch[name] * 1024.0
        wr = write.get(name, 0.0) * 1024.0
        out[short] = {"fetch_bytes_raw": fr, "fetch_bytes_corrected": 2.0 * fr, "write_bytes": wr,
                      "hbm_bytes_per_launch": 2.0 * fr + wr,
                      "note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, KiB units); FETCH_SIZE doubled for "
                              "gfx950 (MI355X_MICROARCH.md, HBM)"}
    return out

if __name__ == "__main__":
    root = sys.argv[1]
    res = {}
    for mode in sys.argv[2:]:
        res.update(main(root, mode))
    print(json.dumps(res, indent=1))
