#!/bin/bash
# tools/build_variant.sh <name> <file.hip> "<extra flags>": tools/bin/libexp_<name>.so = the library with ONE source
# file recompiled with extra flags (the other objects are the regular build's)
set -e
NAME=$1; SRC=$2; FLAGS=$3
cd "$(dirname "$0")/../mcts_q_amd/csrc"
OBJ=../../build/obj
mkdir -p ../../tools/bin ../../build/var
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-slp-vectorize -Wno-unused-function $FLAGS -x hip -c $SRC -o ../../build/var/${NAME}.o
BASE=$(basename $SRC .hip)
OBJS=$(ls $OBJ/*.o | grep -v "/${BASE}.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/bin/libexp_${NAME}.so $OBJS ../../build/var/${NAME}.o
echo built tools/bin/libexp_${NAME}.so
