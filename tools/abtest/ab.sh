#!/bin/bash
# ab.sh -- A/B of alternative builds of librmdf (tools/abtest/*.so, made by build_variant.sh) with bench.py, from the repo root
# on the GPU box.  One script for what used to be run{,2,3,4,5}.sh, run_scene.sh, run_cornell.sh, run_animate.sh, run_flags.sh
# and s_sweep.sh.
#
#   tools/abtest/ab.sh [-s "3 1"] [-c scene[:w:h:max_steps]] [-a "0 0.0167"] [-f flags] [-k steps] [-d] [lib.so ...]
#     -s   frames in flight to run, in this order (default "3 1"; "1 2 3 4 6" = the old s_sweep.sh)
#     -c   scene and size (default 2:1920:1080:256 = the headline; "0:1280:720:128" = the Cornell box, 3 = general power)
#     -a   in_time advance per frame, one run each (default "0"; "0 0.0167 0.1" = the old run_animate.sh)
#     -f   RMDF_FLAGS (default 0; the product knows 4 = raster order, 16 = no pooling, 32 = no Cornell pruning)
#     -k   timed steps per block (default 100)
#     -d   also run the in-tree librmdf.so ("default")
#   libraries default to tools/abtest/*.so
shopt -s nullglob
S="3 1"; CFG="2:1920:1080:256"; AN="0"; FLAGS=0; K=100; DEF=0
while getopts "s:c:a:f:k:d" o; do
  case $o in s) S=$OPTARG;; c) CFG=$OPTARG;; a) AN=$OPTARG;; f) FLAGS=$OPTARG;; k) K=$OPTARG;; d) DEF=1;; *) exit 2;; esac
done
shift $((OPTIND - 1))
IFS=: read -r sc w h ms <<< "$CFG"
w=${w:-1280}; h=${h:-720}; ms=${ms:-128}
libs=("$@"); [ ${#libs[@]} -eq 0 ] && libs=(tools/abtest/*.so)
[ $DEF = 1 ] && libs=(default "${libs[@]}")
export RMDF_FLAGS=$FLAGS
tmp=$(mktemp /tmp/ab.XXXXXX); trap 'rm -f $tmp $tmp.json $tmp.err' EXIT
for s in $S; do for an in $AN; do for lib in "${libs[@]}"; do
  if [ "$lib" = default ]; then unset RMDF_LIB; else export RMDF_LIB=$PWD/$lib; fi
  if python bench.py --scene $sc --width $w --height $h --max-steps $ms --no-cpu-baseline --no-secondary --no-animated \
       --streams $s --steps $K --animate $an 2>$tmp.err > $tmp.json; then
    python tools/show_bench.py $tmp.json | sed "s|$tmp.json|$lib scene=$sc S=$s animate=$an flags=$FLAGS|"
  else
    echo "$lib scene=$sc S=$s flags=$FLAGS: bench.py failed:"; tail -3 $tmp.err
  fi
done; done; done
