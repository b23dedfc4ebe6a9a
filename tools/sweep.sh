#!/bin/bash
# usage: tools/sweep.sh <workload> VAR1=a,b VAR2=c,d ...   -- runs bench.py for the cartesian product of env settings
W=$1; shift
run() {
  local envs="$1"
  line=$(env $envs timeout -k 10 120 python -u bench.py --steps ${STEPS:-50} --warmup 5 --no-cpu-baseline --workload $W 2>/dev/null | grep '^{')
  echo "$line" | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$envs', d['value'], 'ms/step', d['ms_per_step'], 'frac', d['hbm_frac_of_peak'], r['kernel'], r['ms_per_launch'], r['other_kernel']['ms_per_launch'])" 2>/dev/null || echo "$envs FAILED"
}
rec() {
  local prefix="$1"; shift
  if [ $# -eq 0 ]; then run "$prefix"; return; fi
  local var=${1%%=*}; local vals=${1#*=}; shift
  for v in ${vals//,/ }; do rec "$prefix $var=$v" "$@"; done
}
rec "" "$@"
