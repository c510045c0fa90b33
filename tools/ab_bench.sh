#!/bin/bash
# A/B of builds of the library on ONE box (box-to-box spread is +-5 %, larger than most kernel changes):
#   ROUNDS=2 tools/ab_bench.sh trajsde_amd/variants/a.so trajsde_amd/variants/b.so ...
# runs bench.py's forward legs alternately with the in-tree library and with each given one (TRAJSDE_LIB) and prints, per run,
# the 3-stream and 1-stream throughput and the dominant kernel's isolated launch time (HIP events of the one-stream pass).
# FULL=1 also runs the side figures (graph replay, 64x128, SDE step).
rounds=${ROUNDS:-2}
extra="--no-secondary"; [ -n "$FULL" ] && extra=""
mkdir -p gpurun_out
for r in $(seq 1 $rounds); do
  for which in tree "$@"; do
    if [ "$which" = tree ]; then unset TRAJSDE_LIB; else export TRAJSDE_LIB=$PWD/$which; fi
    python3 bench.py --no-cpu-baseline --no-train-step $extra 2>/dev/null | tail -1 | python3 -c "
import json, sys
d = json.loads(sys.stdin.readline())
s1 = d['streams1']
msg = ['%-44s' % '$which', 'scenes/s %.0f' % d['value'], '| 1 stream %.0f (%.3f ms)' % (s1['value'], s1['ms_per_step']),
       '| edge kernel alone %.4f ms frac %.3f' % (d['roofline']['avg_launch_ms'], d['roofline']['frac'])]
if 'config2_64x128' in d: msg.append('| 64x128 %.0f' % d['config2_64x128']['value'])
if isinstance(d.get('roofline_sde_step'), dict) and 'stress_786k' in d['roofline_sde_step']:
    msg.append('| sde_step 786k %.4f ms' % d['roofline_sde_step']['stress_786k']['avg_launch_ms'])
print(*msg)"
  done
done
