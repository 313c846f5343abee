#!/bin/bash
# VGPR / scratch usage of every fused-kernel instantiation: tools/kernel_regs.sh [qnet_fused|qnet_fused_split]
f=${1:-qnet_fused}
cd "$(dirname "$0")/../gnn_hex_amd/csrc"
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -c -Rpass-analysis=kernel-resource-usage -o /tmp/$f.regs.o $f.hip 2>&1 |
  awk '/Function Name:/ {name=$0; sub(/.*Function Name: /,"",name); sub(/ \[.*/,"",name)} / VGPRs:/ {v=$0; sub(/.* VGPRs: /,"",v); sub(/ \[.*/,"",v)} /ScratchSize/ {s=$0; sub(/.*: /,"",s); sub(/ \[.*/,"",s); print name, "vgprs", v, "scratch", s}' | grep qnet_ | sed 's/_ZN6hexgnn15//; s/EEEvNS_8Q.wdArgsE//'
