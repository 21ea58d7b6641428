#!/bin/bash
# A/B of two libraries on the window-kernel shapes: tools/ab_win.sh <base.so> (the tree's library is B)
set -e
base=$1
for args in "27 quad 1 64 20" "27 mag2 1 64 20" "121 mag2 100 64 19" "33 quad 1 16 22"; do
  for lib in $base sdr_amd/libsdrhip.so $base sdr_amd/libsdrhip.so; do
    echo -n "$(basename $lib) "; SDR_LIB_PATH=$PWD/$lib timeout -k 10 120 python tools/exp_win.py $args win
  done
done
