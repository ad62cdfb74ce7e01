#!/bin/bash
# Probe builds: recompile ONE source of libagf_ops.so with extra -D flags and link it against the shipped objects into
# animeface_amd/libagf_ops_<name>.so (git-ignored, travels with gpurun).  tools/time_conv.py etc. pick it up with AGF_PROBE_LIB=<name>.
#   bash tools/probe/build_variant.sh <name> <source.hip> "<flags>"
set -euo pipefail
name=$1; src=$2; flags=${3:-}
cd "$(dirname "$0")/../../animeface_amd/csrc"
mkdir -p build_probe
o=build_probe/${src%.*}_${name}.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=fast -munsafe-fp-atomics -Wall -Wno-unused-function $flags -x hip -c "$src" -o "$o"
objs=""
for f in build/*.o; do
  if [ "$(basename $f)" = "${src%.*}.o" ]; then objs="$objs $o"; else objs="$objs $f"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs -o ../libagf_ops_${name}.so
echo "built libagf_ops_${name}.so"
