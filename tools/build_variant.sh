#!/bin/bash
# tools/build_variant.sh <name> <source.hip> "<-D flags>": libgnxhip with ONE source recompiled with
# extra flags -> tools/_variants/libgnxhip_<name>.so (A/B of two builds on one box: GNX_LIB=...)
set -e
NAME=$1; SRC=$2; FLAGS=$3
ROOT=$(cd "$(dirname "$0")/.." && pwd)
C=$ROOT/geonomics_amd/csrc
mkdir -p $ROOT/tools/_variants/obj
OBJ=$ROOT/tools/_variants/obj/${NAME}_$(basename $SRC .hip).o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-function $FLAGS -c $C/$SRC -o $OBJ
OBJS=""
for f in gnx_api gnx_kernels_pop gnx_kernels_genome gnx_kernels_demog gnx_tile gnx_stats gnx_prim gnx_dd gnx_comm; do
  if [ "$f.hip" == "$SRC" ]; then OBJS="$OBJS $OBJ"; else OBJS="$OBJS $C/_obj/$f.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/tools/_variants/libgnxhip_$NAME.so $OBJS -ldl -lpthread
echo built tools/_variants/libgnxhip_$NAME.so
