# kernel times of the mixed-precision solve at 512^3 under environment settings
cd /tmp && export TMPDIR=/tmp
for e in "$@"; do
  rm -rf /tmp/mp
  env $e timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/mp -- python3 /root/repo/bench.py --workload mixed-3d-512 --repeats 1 --warmup 0 > /tmp/mp.log 2>&1
  echo "== $e"
  python3 - <<'PY'
import csv,glob
f=sorted(glob.glob('/tmp/mp/*/*kernel_stats.csv'))[-1]
for r in csv.DictReader(open(f)):
    if 'f64' in r['Name']:
        print(f"   {r['Name'][7:40]:34s} calls {r['Calls']:>4s}  avg {float(r['AverageNs'])/1e3:9.1f} us")
PY
done
