# plain accesses against the automatic cache policy on grids whose four vectors fit the Infinity Cache (or nearly)
for W in ${@:-2048x2048 4096x2048 4096x3072 4096x4096 5120x4096 6144x4096 4194304 16777216 128x128x128 256x256x256 320x320x320 384x384x384}; do
  echo "#### $W"
  for rep in 1 2; do
  for e in "CG_X=auto" "CG_CACHE_POLICY=0"; do
    echo "== $e"
    env $e timeout -k 10 200 python3 tools/dbg/mini_bench.py $W 200 2>/dev/null | cut -c1-80 || exit 1
  done
  done
done
