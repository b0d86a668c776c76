#!/bin/bash
# dev: build scripts/variants/libbsk_<tag>.so with extra -D flags on ONE translation unit (default k_minimizer_pk).
# usage: scripts/build_variant.sh <tag> "<-D flags>" [unit]
set -e
TAG=$1; FLAGS=$2; UNIT=${3:-k_minimizer_pk}
cd "$(dirname "$0")/../bio_amd/csrc"
mkdir -p ../../scripts/variants
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -I../../include $FLAGS -c -o /tmp/var_$TAG.o $UNIT.hip 2>&1 | grep -v hip-link || true
OBJS=$(ls *.o | grep -v "^$UNIT.o$")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../scripts/variants/libbsk_$TAG.so $OBJS /tmp/var_$TAG.o -lz -ldl -lpthread 2>&1 | grep -v hip-link || true
ls -la ../../scripts/variants/libbsk_$TAG.so
