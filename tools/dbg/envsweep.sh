# usage: envsweep.sh <workload> "ENV1=a ENV2=b" "ENV1=c" ...   -- mini_bench per environment setting, twice each
W=$1; shift
for rep in 1 2; do
for e in "$@"; do
  echo "== $e"
  env $e timeout -k 10 200 python3 tools/dbg/mini_bench.py $W 2>/dev/null
done
done
