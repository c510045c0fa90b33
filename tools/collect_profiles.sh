#!/bin/bash
# Collect the measurements kept under profiles/ (run on the GPU box from the repo root; writes gpurun_out/final/).
#   bash tools/collect_profiles.sh
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/final
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py > $O/bench.log 2>&1
B="$R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-train-step"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks1 -- python3 $B --streams 1 > $O/ks1.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks3 -- python3 $B > $O/ks3.log 2>&1
P="$R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-train-step --streams 1"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmcF -- python3 $P > $O/pmcF.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmcW -- python3 $P > $O/pmcW.log 2>&1
python3 $R/tools/train_step_bench.py > $O/train_step.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kst -- python3 $R/tools/train_step_bench.py --steps 3 --warmup 1 > $O/kst.log 2>&1
cd $R
cp $(find $O/ks1 -name "*kernel_stats.csv" | head -1) $O/kernel_stats_streams1.csv
cp $(find $O/ks3 -name "*kernel_stats.csv" | head -1) $O/kernel_stats_default_streams3.csv
cp $(find $O/kst -name "*kernel_stats.csv" | head -1) $O/train_step_kernel_stats.csv
python3 tools/pmc_summary.py $O/pmcF $O/pmcW $O/pmc_summary.md $O/traffic.json > /dev/null
# the raw traces are large: keep the summaries only
rm -rf $O/ks1 $O/ks3 $O/kst
find $O/pmcF $O/pmcW -name "*kernel_trace.csv" -delete
tail -1 $O/bench.log | cut -c1-300
