# kernel durations and gaps between consecutive kernels of a replayed iteration graph (rocprofv3 kernel trace)
W=${1:-256x256}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/tg
timeout -k 10 120 rocprofv3 --kernel-trace --output-format csv -d /tmp/tg -- python3 /root/repo/tools/dbg/mini_bench.py $W 200 > /tmp/tg.log 2>&1
python3 - <<'PY'
import csv,glob,statistics
f=glob.glob('/tmp/tg/**/*kernel_trace.csv',recursive=True)[0]
rows=[r for r in csv.DictReader(open(f))]
rows.sort(key=lambda r:int(r['Start_Timestamp']))
ks=[(r['Kernel_Name'][:44],int(r['Start_Timestamp']),int(r['End_Timestamp'])) for r in rows]
# steady part: last 400 kernels before the final 50
seg=ks[-460:-60]
dur={}; gap={}
for i in range(1,len(seg)):
    n=seg[i][0]
    dur.setdefault(n,[]).append(seg[i][2]-seg[i][1])
    gap.setdefault(n,[]).append(seg[i][1]-seg[i-1][2])
for n in dur:
    print(f"{n:46s} n={len(dur[n]):4d} duration {statistics.median(dur[n])/1e3:6.2f} us   gap before it {statistics.median(gap[n])/1e3:6.2f} us")
PY
