#!/bin/bash
# kernel trace of a few training steps -> tools/step_timeline.py (gaps between launches), and the torch operators of a step
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r4tl
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $O/kt -- python3 $R/tools/train_step_bench.py --steps 3 --warmup 2 > $O/kt.log 2>&1
python3 $R/tools/step_timeline.py $O/kt --full > $O/timeline.txt 2>&1
python3 $R/tools/train_step_ops.py > $O/ops.txt 2>&1
rm -rf $O/kt
tail -70 $O/timeline.txt
