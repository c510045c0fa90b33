#!/bin/bash
# round-4 probe: GPU parity suite + the recurrence kernel at forced tile counts
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r4_probe2
mkdir -p $O
timeout 900 python3 -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest.log
for tw in 0 1 2 4; do
  if [ $tw = 0 ]; then unset TRAJSDE_RECUR_TW; else export TRAJSDE_RECUR_TW=$tw; fi
  python3 bench.py --no-cpu-baseline --no-train-step --no-secondary --streams 1 --windows 2 --kernel-table > $O/tw$tw.json 2> $O/tw$tw.err
  grep -i "recur" $O/tw$tw.err | head -3
  tail -1 $O/tw$tw.json | cut -c1-160
done
unset TRAJSDE_RECUR_TW
python3 bench.py --no-cpu-baseline --no-train-step --no-secondary --windows 3 | tail -1 | cut -c1-200
