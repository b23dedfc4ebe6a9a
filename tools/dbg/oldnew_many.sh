for W in "$@"; do
  echo "#### $W"
  for lib in tmp_variants/old/conjugate-gradient_amd/lib/libconjugate_gradient_gpu.so conjugate-gradient_amd/lib/libconjugate_gradient_gpu.so; do
    echo -n "$(echo $lib | cut -c1-12): "
    CG_LIB_PATH=$PWD/$lib timeout -k 10 200 python3 tools/dbg/mini_bench.py $W 2>/dev/null
  done
done
