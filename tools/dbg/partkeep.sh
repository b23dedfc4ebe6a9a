# usage: partkeep.sh <workload> MB...   (share of r kept in the Infinity Cache: CG_KEEP_MB against the default)
W=$1; shift
echo "#### $W"
echo "== default"
timeout -k 10 200 python3 tools/dbg/mini_bench.py $W 200 2>/dev/null | cut -c1-80 || exit 1
for mb in "$@"; do
  echo "== CG_KEEP_MB=$mb"
  CG_KEEP_MB=$mb timeout -k 10 200 python3 tools/dbg/mini_bench.py $W 200 2>/dev/null | cut -c1-80 || exit 1
done
