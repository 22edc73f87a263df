#!/bin/bash
# usage: tools/build_variant.sh <name> "<extra flags>" <unit.hip> ["<unit's own flags>"]
# -> variants/lib_<name>.so: the current library with ONE translation unit recompiled under extra -D flags (same-box A/B of
# builds on the GPU: tools/gpu_ab_k.sh).  variants/ is git-ignored and shipped by gpurun.
set -e
cd "$(dirname "$0")/.."
NAME=$1; FLAGS=$2; UNIT=$3; OWN=$4
mkdir -p variants/obj
OBJ=variants/obj/${NAME}_${UNIT}.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 $FLAGS -O3 -std=c++17 -fPIC -Iinclude -Igoi_hyperplane_amd/csrc -munsafe-fp-atomics $OWN -c goi_hyperplane_amd/csrc/$UNIT -o $OBJ
OBJS=$(ls goi_hyperplane_amd/build/*.hip.o | grep -v "/${UNIT}.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o variants/lib_${NAME}.so $OBJS $OBJ
echo built variants/lib_${NAME}.so
