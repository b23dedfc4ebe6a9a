# the same workload with two builds of the library on the same box: usage libcmp.sh <other .so> <workload>...
OTHER=$1; shift
for W in "$@"; do
  echo "#### $W"
  for rep in 1 2; do
    for lib in $OTHER conjugate-gradient_amd/lib/libconjugate_gradient_gpu.so; do
      echo "== $lib"
      CG_LIB_PATH=$PWD/$lib timeout -k 10 200 python3 tools/dbg/mini_bench.py $W 2>/dev/null | cut -c1-80 || exit 1
    done
  done
done
