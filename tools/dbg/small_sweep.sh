#!/bin/bash
# single-workgroup solver against the two-kernel iteration on grids of 8193..16384 points
# usage: small_sweep.sh [steps] [grids...]   (a grid that converges inside the timed regions reads as 0 ms)
steps=${1:-300}
shift
grids=${@:-"16384 128x128 120x120 25x25x25 11x11x11x11 128x128:2 16384:2"}
out=gpurun_out/small_sweep.log
: > $out
for g in $grids; do
  for sm in 1 0; do
    echo "== $g CG_SMALL=$sm" >> $out
    CG_SMALL=$sm timeout -k 10 120 python tools/dbg/mini_bench.py $g $steps 2>&1 | grep -v amdgpu.ids >> $out || exit 1
  done
done
cat $out
