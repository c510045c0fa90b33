#!/bin/bash
# block-size sweep on the GPU box: prints ms/step and the per-kernel table for each setting
cd "$(dirname "$0")/.."
run() {
  echo "=== $*"
  env "$@" python bench.py --steps 10 --warmup 3 --no-cpu-baseline --kernel-table 2>&1 | grep -E "ms/fwd|ms_per_step|total" | sed -E 's/.*"ms_per_step": ([0-9.]+).*"frac": ([0-9.]+).*/ms_per_step \1 frac \2/' | head -12
}
run TRAJSDE_THREADS_EDGE=512
run TRAJSDE_THREADS_EDGE=768
run TRAJSDE_THREADS_EDGE=1024
run TRAJSDE_THREADS_DECODE=512
run TRAJSDE_THREADS_DECODE=640
run TRAJSDE_THREADS_RECUR=128
run TRAJSDE_THREADS_RECUR=512
run TRAJSDE_THREADS_NODE=256
run TRAJSDE_THREADS_NODE=1024
