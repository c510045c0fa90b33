#!/bin/bash
# kernel-trace stats of one kernel under several builds of the library (TRAJSDE_LIB), for on-box experiments:
#   bash tools/variant_stats.sh <kernel-name-substring> lib1.so lib2.so ...
pat=$1; shift
for lib in "" "$@"; do
  if [ -n "$lib" ]; then export TRAJSDE_LIB=$PWD/$lib; else unset TRAJSDE_LIB; fi
  bash tools/quick_kernel_stats.sh 40 | grep -- "$pat" | sed "s|^|${lib:-in-tree}: |"
done
