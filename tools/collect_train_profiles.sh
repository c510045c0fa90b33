#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r4train
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python3 $R/tools/train_step_bench.py > $O/train_step.log 2>&1
python3 $R/tools/train_step_bench.py --pipeline > $O/train_step_pipelined.log 2>&1
python3 $R/tools/train_step_bench.py --config config4 > $O/train_step_config4.log 2>&1
TRAJSDE_WGRAD_F32=1 TRAJSDE_IMMEDIATE_SUMS=1 TRAJSDE_RECUR_LEGACY=1 python3 $R/tools/train_step_bench.py > $O/train_step_round_start_forms.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kst -- python3 $R/tools/train_step_bench.py --steps 3 --warmup 1 > $O/kst.log 2>&1
cp $(find $O/kst -name "*kernel_stats.csv" | head -1) $O/train_step_kernel_stats.csv
rm -rf $O/kst
tail -3 $O/train_step.log | cut -c1-200
