#!/bin/bash
# build_variant.sh NAME [extra hipcc flags...]  ->  build/variants/libsdfr_NAME.so  (timing experiments)
# The product's Makefile with another output, object directory and extra flags (e.g. -DSDFR_FWD_WAVES=4).
set -e
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
NAME=$1; shift
mkdir -p "$ROOT/build/variants"
make -s -C "$ROOT/sdfest_amd/csrc" OUT="$ROOT/build/variants/libsdfr_$NAME.so" OBJDIR="$ROOT/build/obj_$NAME" EXTRA="$*" > /dev/null
echo "built build/variants/libsdfr_$NAME.so"
