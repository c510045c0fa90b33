#!/bin/bash
# Build another copy of the library with extra compiler flags, for on-box A/B runs (tools/ab_bench.sh, TRAJSDE_LIB):
#   tools/build_variant.sh <name> "<extra hipcc flags>"      ->  trajsde_amd/variants/<name>.so
# Objects go to a scratch directory; the in-tree library is not touched.  *.so is git-ignored but travels with gpurun.
set -e
name=$1; flags=$2
root=$(cd "$(dirname "$0")/.." && pwd)
obj=/tmp/trajsde_variant_$name; mkdir -p $obj $root/trajsde_amd/variants
pids=()
for src in $root/trajsde_amd/csrc/*.hip; do
  b=$(basename $src .hip)
  extra=""; [ $b = recur ] && extra="-mllvm -amdgpu-mfma-vgpr-form"      # trajsde_amd/build.py PER_FILE_FLAGS
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize -DTSDE_NO_SLP=1 -std=c++17 -fPIC -Wall -Wno-unused-function $extra $flags -c $src -o $obj/$b.o &
  pids+=($!)
done
for p in "${pids[@]}"; do wait $p; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $root/trajsde_amd/variants/$name.so $obj/*.o
echo "built trajsde_amd/variants/$name.so"
