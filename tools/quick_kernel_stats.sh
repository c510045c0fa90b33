#!/bin/bash
# kernel-trace stats of the one-stream forward (top kernels), for a quick look after a kernel change:
#   bash tools/quick_kernel_stats.sh [n_rows]      (on the GPU box; writes gpurun_out/quick/)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/quick
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks -- python3 $R/bench.py --steps 20 --warmup 3 --windows 1 --no-cpu-baseline --no-train-step --no-secondary --streams 1 > $O/ks.log 2>&1
cd $R
f=$(find $O/ks -name "*kernel_stats.csv" | head -1)
cp $f $O/kernel_stats.csv
python3 - "$O/kernel_stats.csv" "${1:-16}" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:int(sys.argv[2])]:
    print(f"{r['Name'][:70]:70s} {r['Calls']:>6s} {float(r['AverageNs'])/1e3:9.1f} us {100*float(r['TotalDurationNs'])/tot:5.1f}%")
PY
