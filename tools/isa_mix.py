#!/usr/bin/env python3
"""Static instruction mix of the layer loop of a fused kernel (round 4: VALU instructions cost fp32-MFMA time).
   tools/isa_mix.py [NT]   -> compiles qnet_fwd/bwd_kernel<NT, 0> alone to ISA and prints, per kernel, the histogram of
   the instructions between the loop header that contains the MFMAs and its back edge, plus VGPR / scratch figures."""
import collections, os, re, subprocess, sys, tempfile
nt = sys.argv[1] if len(sys.argv) > 1 else "7"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tmp = tempfile.mkdtemp()
src = os.path.join(tmp, "one.hip")
open(src, "w").write('#include "qnet_fused_kernels.h"\nnamespace hexgnn {\n'
                     'template __global__ void qnet_fwd_kernel<%s, 0>(QFwdArgs);\n'
                     'template __global__ void qnet_bwd_kernel<%s, 0>(QBwdArgs);\n}\n' % (nt, nt))
out = os.path.join(tmp, "one.s")
r = subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-Wno-unused-function", "-S",
                    "--cuda-device-only", "-I", os.path.join(root, "gnn_hex_amd", "csrc"), "-Rpass-analysis=kernel-resource-usage",
                    "-o", out, src] + sys.argv[2:], capture_output=True, text=True)
for line in r.stderr.splitlines():
    if re.search(r"Function Name|VGPRs:|ScratchSize|error", line):
        print(re.sub(r".*remark: [^ ]* ", "", line).replace("[-Rpass-analysis=kernel-resource-usage]", "").strip())
text = open(out).read().splitlines()
print("ISA:", out)
starts = [i for i, l in enumerate(text) if re.match(r"^_ZN6hexgnn15qnet_(fwd|bwd)_kernel.*:", l)]
for s in starts:
    e = next(i for i in range(s, len(text)) if "s_endpgm" in text[i])
    body = text[s:e]
    # the layer loop: from the loop header with the most MFMAs inside to the last line that branches back to it
    heads = [i for i, l in enumerate(body) if "Loop Header: Depth=1" in l]
    best = None
    for h in heads:
        label = body[h].split(":")[0]
        backs = [i for i, l in enumerate(body) if re.search(r"s_c?branch\S*\s+%s\b" % re.escape(label), l)]
        if not backs: continue
        # out-of-line blocks of the loop may sit behind the back edge: take every line tagged with this loop header
        tag = "Header=%s " % label.lstrip(".L")
        lines = [l for l in body[h:] if True]
        inloop, cur = [], True
        blocks = []
        # walk basic blocks: a block belongs to the loop if its label line carries the tag (or it is the header block)
        keep = True
        for l in body:
            if re.match(r"^\.LBB", l) or l.startswith("; %bb."):
                keep = ("in Loop: Header=%s" % label.lstrip(".L")) in l or l.startswith(label)
            if keep: inloop.append(l)
        n_mfma = sum("v_mfma" in l for l in inloop)
        if best is None or n_mfma > best[0]: best = (n_mfma, inloop)
    hist = collections.Counter()
    for l in best[1]:
        m = re.match(r"^\s+([a-z][a-z0-9_]+)", l)
        if m: hist[re.sub(r"_e32|_e64|_sdwa|_dpp", "", m.group(1))] += 1
    valu = sum(c for k, c in hist.items() if k.startswith("v_") and "mfma" not in k)
    print("\n%s\n  loop: %d MFMA, %d other VALU, %d LDS, %d SALU, %d VMEM" % (
        body[0].split(":")[0][:40], best[0], valu, sum(c for k, c in hist.items() if k.startswith("ds_")),
        sum(c for k, c in hist.items() if k.startswith("s_")), sum(c for k, c in hist.items() if k.startswith(("buffer_", "global_")))))
    print("  " + ", ".join("%s %d" % kv for kv in hist.most_common(24)))
