# the same benchmark line with last round's library and with the current one, on the same box
W=${1:-128x128x128x128}
for lib in tmp_variants/old/conjugate-gradient_amd/lib/libconjugate_gradient_gpu.so conjugate-gradient_amd/lib/libconjugate_gradient_gpu.so; do
  for rep in 1 2; do
  echo "== $lib"
  CG_LIB_PATH=$PWD/$lib timeout -k 10 200 python3 tools/dbg/mini_bench.py $W
  done
done
