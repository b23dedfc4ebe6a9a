# kernel A / B times of the 4-D 128^4 grid for different slot patches (CG_COLS_T2)
for t in 1 2 4; do
  echo "== CG_COLS_T2=$t"
  CG_COLS_T2=$t timeout -k 10 200 python3 bench.py --gpus 1 --steps 40 --warmup 6 --repeats 3 --no-cpu-baseline --no-traffic --workload 128x128x128x128 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); r=d['roofline']; print(d['value'], d['ms_per_step'], r['kernel'], r['ms_per_launch'], r['other_kernel']['ms_per_launch'])"
done
