#!/bin/bash
# round-4 probe: the recurrence kernel at forced tile counts per workgroup (TRAJSDE_RECUR_TW), baseline forward, training step
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r4_probe1
mkdir -p $O
for tw in 0 1 2 4; do
  if [ $tw = 0 ]; then unset TRAJSDE_RECUR_TW; else export TRAJSDE_RECUR_TW=$tw; fi
  python3 bench.py --no-cpu-baseline --no-train-step --no-secondary --streams 1 --windows 2 --kernel-table > $O/tw$tw.json 2> $O/tw$tw.err
  grep -i "recur\|k_enc" $O/tw$tw.err | head -5
  tail -1 $O/tw$tw.json | cut -c1-200
done
unset TRAJSDE_RECUR_TW
python3 tools/train_step_bench.py > $O/train.log 2>&1
tail -40 $O/train.log
