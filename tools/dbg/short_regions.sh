run() { echo "== $*"; env "$@" timeout -k 10 200 python3 bench.py --gpus 1 --steps 20 --warmup 5 --repeats 7 --no-cpu-baseline --no-traffic --no-beyond-cache $EXTRA 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(d['value'], d['ms_per_step'], d['timing']['ms_per_step_all'])"; }
EXTRA="" run A=1
EXTRA="--no-graph" run A=1
EXTRA="" run CG_GRAPH_ITERS=10
EXTRA="" run CG_GRAPH_ITERS=4
