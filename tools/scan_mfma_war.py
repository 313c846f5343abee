#!/usr/bin/env python
"""Scan the fused kernels' ISA for the pattern behind the width-24 bug: an MFMA whose accumulator
moves (vdst != srcC) followed closely by an LDS / vector-memory load that writes the dead srcC registers.  Nothing
interlocks the load's return against the MFMA's pending srcC read when the MFMA still waits in the matrix pipe.

    python tools/scan_mfma_war.py [qnet_fused|qnet_fused_split] [window]
"""
import re, subprocess, sys, os
tu = sys.argv[1] if len(sys.argv) > 1 else "qnet_fused"
win = int(sys.argv[2]) if len(sys.argv) > 2 else 16
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
asm = "/tmp/%s.scan.s" % tu
subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-Wno-unused-function", "-S",
                "--cuda-device-only", "-o", asm, os.path.join(root, "gnn_hex_amd/csrc", tu + ".hip")],
               stderr=subprocess.DEVNULL, check=True)
def rng(s):
    m = re.match(r"v\[(\d+):(\d+)\]", s)
    if m: return int(m.group(1)), int(m.group(2))
    m = re.match(r"v(\d+)$", s)
    if m: return int(m.group(1)), int(m.group(1))
    return None
cur, lines = None, []
hits = {}
for ln in open(asm):
    m = re.match(r"^(_ZN6hexgnn\S+?):", ln)
    if m: cur, lines = m.group(1), []
    t = ln.strip()
    if not t or t.startswith(";") or t.startswith("."): continue
    lines.append(t)
    m = re.match(r"(ds_read\S*|buffer_load\S*|global_load_dword\S*)\s+(v\[\d+:\d+\]|v\d+),", t)
    if not m or cur is None: continue
    d = rng(m.group(2))
    for back, prev in enumerate(reversed(lines[-win - 1:-1])):
        mm = re.match(r"v_mfma_\S+\s+(v\[\d+:\d+\]),\s*\S+,\s*\S+,\s*(v\[\d+:\d+\])", prev)
        if not mm: continue
        vd, vc = rng(mm.group(1)), rng(mm.group(2))
        if vd != vc and not (d[1] < vc[0] or d[0] > vc[1]):
            # how many MFMAs earlier was this srcC produced?  (the MFMA waits inside the matrix pipe -- and reads srcC late,
            # the only way the load can overtake it -- only when its producer is closer than the dependent latency:
            # 40 cycles for v_mfma_f32_16x16x4_f32 at 32 cycles per issue, i.e. a distance of 1 or 2 slots)
            idx = len(lines) - 2 - back
            dist, k = None, 0
            for pp in reversed(lines[:idx]):
                m2 = re.match(r"v_mfma_\S+\s+(v\[\d+:\d+\]),", pp)
                if m2:
                    k += 1
                    pd = rng(m2.group(1))
                    if not (pd[1] < vc[0] or pd[0] > vc[1]):
                        dist = k
                        break
                elif re.match(r"\S+\s+(v\[\d+:\d+\]|v\d+),", pp):
                    wd = rng(re.match(r"\S+\s+(v\[\d+:\d+\]|v\d+),", pp).group(1))
                    if wd and not (wd[1] < vc[0] or wd[0] > vc[1]):
                        dist = 99           # srcC written by a non-MFMA instruction (initialisation / move): complete at issue
                        break
                if k > 12: break
            hits.setdefault(cur, []).append((back + 1, prev, t, dist if dist is not None else 99))
tot_close = 0
for k, v in hits.items():
    short = re.sub(r"^_ZN6hexgnn\d+", "", k)[:44]
    close = [x for x in v if x[3] <= 2]
    tot_close += len(close)
    print("%s: %d load(s) onto a moved accumulator's srcC within %d instructions (nearest %d); %d of them behind an MFMA whose "
          "srcC was produced <= 2 MFMA slots earlier" % (short, len(v), win, min(x[0] for x in v), len(close)))
    ex = close[0] if close else v[0]
    print("    e.g.", ex[1], " ...  ", ex[2], " (producer %s slots back)" % ex[3])
if not hits: print("no such pattern within %d instructions" % win)
print("TOTAL sites with a dependent distance <= 2 slots:", tot_close)
