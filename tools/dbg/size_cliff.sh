# per-point cost of the 2-D iteration around the Infinity Cache's capacity (rows of 8192 points)
for n1 in 4096 5120 6144 7168 7680 8192 8704 9216 10240 12288; do
  echo "== 8192x$n1"
  timeout -k 10 200 python3 tools/dbg/mini_bench.py 8192x$n1 2>/dev/null | cut -c1-80 || exit 1
done
