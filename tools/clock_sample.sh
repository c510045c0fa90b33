#!/bin/bash
# sample the GPU's clock and power while the forward runs (is the dominant kernel power-limited?):
#   bash tools/clock_sample.sh      -> gpurun_out/clock_sample.log
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
python3 $R/bench.py --steps 1500 --warmup 3 --windows 3 --no-cpu-baseline --no-train-step --no-secondary --streams 1 > $R/gpurun_out/clock_bench.log 2>&1 &
pid=$!
for i in $(seq 1 90); do
  rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power|fclk|mclk" | tr '\n' ' '
  echo
  sleep 0.3
done > $R/gpurun_out/clock_sample.log
wait $pid
tail -c 300 $R/gpurun_out/clock_bench.log
