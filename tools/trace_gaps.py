"""Idle gaps of the GPU timeline from a rocprofv3 --kernel-trace CSV: per kernel name the average duration and the average
idle time BEFORE it starts (start - previous kernel's end), over the steady part of the run.
    python tools/trace_gaps.py <..._kernel_trace.csv> [skip_first_fraction]"""
import collections
import csv
import sys


def main(path, skip=0.3):
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][-60:]))
    rows.sort()
    rows = rows[int(len(rows) * skip):]
    dur, gap, cnt = collections.defaultdict(float), collections.defaultdict(float), collections.defaultdict(int)
    prev_end = None
    for s, e, name in rows:
        dur[name] += e - s
        if prev_end is not None:
            gap[name] += max(0, s - prev_end)
        cnt[name] += 1
        prev_end = max(prev_end or e, e)
    span = rows[-1][1] - rows[0][0]
    busy = sum(dur.values())
    print("span %.3f ms, busy %.3f ms (%.1f %%), %d launches" % (span / 1e6, busy / 1e6, 100.0 * busy / span, len(rows)))
    print("%-62s %6s %10s %12s" % ("kernel", "calls", "avg us", "idle before us"))
    for name in sorted(cnt, key=lambda k: -dur[k]):
        print("%-62s %6d %10.1f %12.1f" % (name, cnt[name], dur[name] / cnt[name] / 1e3, gap[name] / cnt[name] / 1e3))


if __name__ == "__main__":
    main(sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 0.3)
