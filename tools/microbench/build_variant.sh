#!/bin/bash
# build_variant.sh NAME [extra hipcc flags...]  ->  build/variants/libsdfr_NAME.so  (timing experiments)
set -e
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
NAME=$1; shift
mkdir -p "$ROOT/build/variants"
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -shared -I"$ROOT/include" -Wall -Wno-unused-function \
  -fvisibility=hidden "$@" -x hip "$ROOT"/sdfest_amd/csrc/*.hip "$ROOT"/sdfest_amd/csrc/api.cpp \
  -o "$ROOT/build/variants/libsdfr_$NAME.so"
echo "built build/variants/libsdfr_$NAME.so"
