#!/bin/bash
# Build a variant of the library for a same-box A/B (scripts/ab_variants.sh):
#   scripts/build_variant.sh <name> [extra hipcc flags, e.g. -DSOME_EXPERIMENT_MACRO]
# -> scripts/variants/<name>.so (git-ignored; travels with gpurun)
set -e
cd "$(dirname "$0")/.."
name=$1; shift
mkdir -p scripts/variants /tmp/binf_variant_$name
objs=()
for src in binf_amd/csrc/*.hip; do
  extra=""
  [ "$(basename $src)" = "poly.hip" ] && extra="-mllvm -amdgpu-mfma-vgpr-form"
  obj=/tmp/binf_variant_$name/$(basename ${src%.hip}).o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-function $extra "$@" -c $src -o $obj &
  objs+=($obj)
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o scripts/variants/$name.so "${objs[@]}"
rm -rf /tmp/binf_variant_$name
ls -la scripts/variants/$name.so
