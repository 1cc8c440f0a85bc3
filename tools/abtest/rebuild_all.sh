#!/bin/bash
# rebuild_all.sh -- the A/B builds of the CURRENT product sources that tools/gpu_measure.sh times (tools/abtest/<name>.so, git-ignored, they
# travel to the GPU box with the snapshot).  r04.so (round 4's sources) is not rebuilt: `git archive a05000d` + build_variant.sh there.
cd "$(dirname "$0")" || exit 1
./build_variant.sh noxl     -DRMDF_AB_NO_XL &
./build_variant.sh xl4      -DRMDF_AB_XL_G=4 &
./build_variant.sh sharedb  -DRMDF_AB_SHARED_BOUNDS &
./build_variant.sh mirror16 -DRMDF_AB_MIRROR16 &
./build_variant.sh mt48     -DRMDF_AB_MERGE_T=48 &
./build_variant.sh mt56     -DRMDF_AB_MERGE_T=56 &
wait
ls -la *.so
