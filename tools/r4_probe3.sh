#!/bin/bash
# recurrence backward at forced tile counts + training forward; backward parity
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r4_probe3
mkdir -p $O
timeout 900 python3 -m pytest tests/test_gpu_backward.py -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $O/pytest.log
for tw in 0 1 2 3; do
  if [ $tw = 0 ]; then unset TRAJSDE_RECUR_BWD_TW; else export TRAJSDE_RECUR_BWD_TW=$tw; fi
  python3 tools/train_step_bench.py > $O/train_tw$tw.log 2>&1
  echo "bwd tw=$tw"; grep "ms_per_train_step" $O/train_tw$tw.log | cut -c1-120; grep "recur" $O/train_tw$tw.log
done
