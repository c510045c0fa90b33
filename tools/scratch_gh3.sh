export TRAJSDE_REL_SPLIT=1
for lib in tree h3occ2 h3exp1 h3exp2 h3exp3; do
  if [ $lib = tree ]; then unset TRAJSDE_LIB; else export TRAJSDE_LIB=$PWD/trajsde_amd/variants/$lib.so; fi
  echo "== $lib"
  bash tools/quick_kernel_stats.sh 40 | grep "k_global_attn"
done
