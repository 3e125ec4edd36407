#!/bin/bash
# A variant of librnad_hip.so with one source recompiled under extra flags (kernel A/B experiments; RNAD_HIP_SO selects it at run time):
#   tools/build_variant.sh <name> <source.hip> <extra hipcc flags...>   ->  r-nad_amd/csrc/_variants/<name>.so
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
cd $R/r-nad_amd/csrc
name=$1; src=$2; shift 2
mkdir -p _variants/_obj_$name
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -fvisibility=hidden -I../../include -Wall -Wno-unused-function"
EXTRA=""
case $src in
  mlp_fwd.hip|mlp_bwd_t.hip|mlp_rows.hip) EXTRA="-mllvm -amdgpu-mfma-vgpr-form -fno-honor-nans";;
  mlp_bwd.hip) EXTRA="-fno-honor-nans";;
esac
/opt/rocm/bin/hipcc $FLAGS $EXTRA "$@" -c $src -o _variants/_obj_$name/${src%.*}.o
objs=""
for o in _obj/*.o; do
  b=$(basename $o)
  if [ "$b" == "${src%.*}.o" ]; then objs="$objs _variants/_obj_$name/$b"; else objs="$objs $o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs -o _variants/$name.so
echo built _variants/$name.so
