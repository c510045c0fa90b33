#!/bin/bash
# Round 6: everything kept under profiles/r06_* in one go (run on the GPU box from the repo root; writes gpurun_out/r06/).
#   bash tools/collect_r06.sh
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
export ROUND=r06
O=$R/gpurun_out/r06
mkdir -p $O
bash $R/tools/collect_counters.sh > $O/collect_counters.log 2>&1      # kernel stats (1 / 3 streams, SDE step, stress), SQ / TCC counters, traffic
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py > $O/bench.log 2> $O/bench.err
python3 $R/tools/train_step_bench.py --pipeline > $O/train_step_pipelined.log 2>&1
python3 $R/tools/train_step_bench.py --pipeline --config config4 > $O/train_step_config4_pipelined.log 2>&1
python3 $R/tools/train_step_bench.py --config config1 --steps 20 > $O/train_step_host_floor.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kstr -- python3 $R/tools/train_step_bench.py --steps 3 --warmup 1 > $O/kstr.log 2>&1
cp $(find $O/kstr -name "*kernel_stats.csv" | head -1) $O/train_step_kernel_stats.csv
rm -rf $O/kstr
rocprofv3 --kernel-trace --output-format csv -d $O/kt -- python3 $R/tools/train_step_bench.py --steps 3 --warmup 2 > $O/kt.log 2>&1
python3 $R/tools/step_timeline.py $O/kt --full > $O/train_step_timeline.txt 2>&1      # every launch of one step in start order
rm -rf $O/kt
python3 $R/tools/train_loop_stalls.py > $O/train_loop_stalls.log 2>/dev/null
python3 $R/tools/graph_stage_bench.py > $O/graph_stage.log 2>&1
python3 $R/tools/graph_stage_bench.py --exact >> $O/graph_stage.log 2>&1
python3 $R/tools/graph_stage_bench.py --workload config2 >> $O/graph_stage.log 2>&1
# k_sde_step under a warmed clock: HIP events of 2000 launches after 500 warm-up launches, three times, and the profiler's view of the
# same command (the three figures published in round 4 -- 0.28 / 0.304 / 0.311 -- were three commands on three boxes)
for i in 1 2 3; do python3 $R/tools/sde_step_bench.py 786432 2000 >> $O/sde_step_warm.log 2>&1; done
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kssw -- python3 $R/tools/sde_step_bench.py 786432 2000 >> $O/sde_step_warm.log 2>&1
cp $(find $O/kssw -name "*kernel_stats.csv" | head -1) $O/kernel_stats_sde_step_warm.csv
rm -rf $O/kssw
cd $R
tail -1 $O/bench.log | cut -c1-200
