#!/bin/bash
# build_variant.sh <name> [-DKNOB | -mllvm ... ...]: another build of the product sources as tools/abtest/<name>.so (A/B with ab.sh).
# Goes through csrc/Makefile (objects under csrc/build/ab/<name>), so a variant differs from the product by the given flags only --
# they are appended to every compile, after the per-scene scheduler flags.
here="$(cd "$(dirname "$0")" && pwd)"
name=$1; shift
make -s -C "$here/../../ray-marching-distance-fields_amd/csrc" -j4 OBJDIR=build/ab/$name OUT="$here/$name.so" EXTRA="$*" "$here/$name.so"
