#!/bin/bash
# usage: ab_env2.sh "ENV1=.. ENV2=.." label
for r in 1 2; do
for which in default "$1"; do
  if [ "$which" = default ]; then pre=""; else pre="$1"; fi
  env $pre python3 bench.py --no-cpu-baseline --no-train-step --no-secondary 2>/dev/null | tail -1 | python3 -c "
import json, sys
d = json.loads(sys.stdin.readline())
s1 = d['streams1']
print('%-60s' % '$which', 'scenes/s %.0f' % d['value'], '| 1 stream %.0f (%.3f ms)' % (s1['value'], s1['ms_per_step']), '| edge kernel alone %.4f ms frac %.3f' % (d['roofline']['avg_launch_ms'], d['roofline']['frac']))"
done
done
