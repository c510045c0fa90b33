#!/bin/bash
# A/B of one environment switch of the library on ONE box:  tools/ab_env.sh VAR=value [rounds]
# runs bench.py's forward legs alternately without and with the assignment
kv=$1; rounds=${2:-2}
for r in $(seq 1 $rounds); do
  for which in default "$kv"; do
    if [ "$which" = default ]; then pre=""; else pre="$kv"; fi
    env $pre python3 bench.py --no-cpu-baseline --no-train-step 2>/dev/null | tail -1 | python3 -c "
import json, sys
d = json.loads(sys.stdin.readline())
s1 = d['streams1']
print('$which', 'scenes/s %.0f' % d['value'], '| 1 stream %.0f (%.3f ms)' % (s1['value'], s1['ms_per_step']),
      '| edge kernel alone %.4f ms frac %.3f' % (d['roofline']['avg_launch_ms'], d['roofline']['frac']),
      '| graph %.0f' % d['graph_replay']['value'], '| 64x128 %.0f' % d['config2_64x128']['value'])"
  done
done
