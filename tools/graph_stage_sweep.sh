#!/bin/bash
# graph-stage rows of the per-kernel table (bench.py --kernel-table) for a few step-group sizes of the snapshot kernels
for tg in ${1:-7 5 3}; do
  echo "== TRAJSDE_AA_TG=$tg"
  TRAJSDE_AA_TG=$tg python bench.py --no-cpu-baseline --no-train-step --no-secondary --kernel-table --steps 10 --windows 2 2>&1 >/dev/null | grep -E "k_prep_first|k_scatter2|k_row_sort2|k_aa_count|k_scan_multi|k_graph_fill|k_collect|total"
done
