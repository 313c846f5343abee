#!/bin/bash
# A/B of builds of libhexgnn.so on ONE box (gpurun): tools/ab.sh ab/libhexgnn_base.so ab/libhexgnn_exp.so [more libs ...]
# Alternates the libraries REPS (default 3) times and prints value / ms_per_step / the three kernel times of each run.
# Extra bench arguments: BENCH_ARGS="--data D1" tools/ab.sh ...
for rep in $(seq 1 ${REPS:-3}); do
  for lib in "$@"; do
    cp $lib gnn_hex_amd/libhexgnn.so
    python bench.py --no-cpu-baseline --no-split --no-other-configs --sustain-s 0 --steps 100 --warmup 20 $BENCH_ARGS 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['roofline']['kernel_ms_per_step']
print('$lib', round(d['value']), round(d['ms_per_step'],4), {n[:8]: round(v*1e3,1) for n,v in k.items()})"
  done
done
cp $1 gnn_hex_amd/libhexgnn.so
