#!/bin/bash
# A/B of two builds of the library on ONE box (box-to-box spread is +-5 %, larger than most kernel changes):
#   tools/ab_bench.sh trajsde_amd/libtrajsde_hip_base.so [rounds]
# runs bench.py's forward legs alternately with the in-tree library and with the given one (TRAJSDE_LIB) and prints, per run,
# the 3-stream and 1-stream throughput and the dominant kernel's isolated launch time.
other=$1; rounds=${2:-2}
mkdir -p gpurun_out
for r in $(seq 1 $rounds); do
  for which in tree other; do
    if [ $which = other ]; then export TRAJSDE_LIB=$PWD/$other; else unset TRAJSDE_LIB; fi
    python3 bench.py --no-cpu-baseline --no-train-step 2>/dev/null | tail -1 | python3 -c "
import json, sys
d = json.loads(sys.stdin.readline())
s1 = d['streams1']
print('$which', 'scenes/s %.0f' % d['value'], '| 1 stream %.0f (%.3f ms)' % (s1['value'], s1['ms_per_step']),
      '| edge kernel alone %.4f ms frac %.3f' % (s1['roofline']['avg_launch_ms'], s1['roofline']['frac']),
      '| 64x128 %.0f' % d['config2_64x128']['value'])"
  done
done
