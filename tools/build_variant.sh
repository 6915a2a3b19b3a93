#!/bin/bash
# usage: tools/build_variant.sh <name> <file.hip> [extra hipcc flags...]   (build container, repo root)
# Recompiles ONE translation unit with extra flags (e.g. -DCGE_TRAFFIC_WAVES=5) and links it with the objects of the last full
# build into tools/ab/libcge_<name>.so — the A/B library CGE_AMD_LIBRARY points bench.py / the tests at (tools/ab_bench.sh).
set -e
NAME=$1; SRC=$2; shift 2
PKG=custom_gymnasium_environments_amd
mkdir -p tools/ab $PKG/build/var_$NAME
OBJ=$PKG/build/var_$NAME/$(basename $SRC).o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -Wno-unused-local-typedef -fno-fast-math \
  -ffp-contract=off -fno-strict-aliasing "$@" -c $PKG/csrc/$SRC -o $OBJ
OTHERS=$(ls $PKG/build/*.hip.o | grep -v "/$(basename $SRC).o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJ $OTHERS -o tools/ab/libcge_$NAME.so
echo tools/ab/libcge_$NAME.so
