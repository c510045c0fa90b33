#!/bin/bash
# global attention on the matrix cores: parity, then the one-stream kernel table with it on and off
python3 -m pytest tests/test_gpu_parity.py -x -q 2>&1 | tail -3
for v in 1 0; do
  TRAJSDE_LIB=$PWD/trajsde_amd/variants/libtrajsde_alt.so TRAJSDE_GATTN_MM=$v python3 bench.py --no-cpu-baseline --no-train-step --no-secondary --streams 1 --windows 2 --kernel-table 2>&1 >/tmp/o.json | grep -i "global_attn\|total"
  tail -1 /tmp/o.json | cut -c1-150
done
