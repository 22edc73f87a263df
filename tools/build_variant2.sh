#!/bin/bash
# usage: tools/build_variant2.sh <name> "<extra flags>"  -> variants/lib_<name>.so with reduce_rows.hip AND api.hip rebuilt under the flags
set -e
cd "$(dirname "$0")/.."
NAME=$1; FLAGS=$2
mkdir -p variants/obj
for U in reduce_rows.hip api.hip; do
/opt/rocm/bin/hipcc --offload-arch=gfx950 $FLAGS -O3 -std=c++17 -fPIC -Iinclude -Igoi_hyperplane_amd/csrc -munsafe-fp-atomics -c goi_hyperplane_amd/csrc/$U -o variants/obj/${NAME}_$U.o &
done; wait
OBJS=$(ls goi_hyperplane_amd/build/*.hip.o | grep -v "/reduce_rows.hip.o" | grep -v "/api.hip.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o variants/lib_${NAME}.so $OBJS variants/obj/${NAME}_reduce_rows.hip.o variants/obj/${NAME}_api.hip.o
echo built variants/lib_${NAME}.so
