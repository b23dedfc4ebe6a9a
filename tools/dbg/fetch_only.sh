# FETCH_SIZE of kernel A under an environment setting: usage fetch_only.sh <workload> "ENV=.."...
W=$1; shift
cd /tmp && export TMPDIR=/tmp
for e in "$@"; do
  rm -rf /tmp/fo; 
  env $e timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/fo -- python3 /root/repo/bench.py --pmc-child --workload $W > /tmp/fo.log 2>&1
  python3 - "$e" <<'PY'
import csv,glob,sys,statistics
f=glob.glob('/tmp/fo/**/*counter_collection.csv',recursive=True)
per={}
for r in csv.DictReader(open(f[0])):
    n=r['Kernel_Name']
    if 'k_cols' in n or 'k_planes' in n or 'k_rows' in n:
        per.setdefault(n[:60],[]).append(float(r['Counter_Value']))
for k,v in per.items():
    full=[x for x in v if x>0.2*max(v)]
    print(sys.argv[1], k, 'fetch GB', round(statistics.mean(full)*1024*2/1e9,3))
PY
done
