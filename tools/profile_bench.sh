#!/bin/bash
# Profiles `bench.py` on the GPU box: one kernel-trace/stats pass and two counter passes
# (FETCH_SIZE and WRITE_SIZE need separate passes: 3 + 2 of the 4 TCC slots).  Results land
# in gpurun_out/prof_<tag>/; tools/summarize_profile.py turns them into profiles/<tag>_*.
# usage: tools/profile_bench.sh <tag> [bench args...]
set -e
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $ROOT/bench.py --no-cpu-baseline --no-traffic --no-beyond-cache "$@" > $OUT/stats.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -- python3 $ROOT/bench.py --steps 20 --warmup 2 --repeats 1 --no-cpu-baseline --no-traffic --no-beyond-cache "$@" > $OUT/fetch.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -- python3 $ROOT/bench.py --steps 20 --warmup 2 --repeats 1 --no-cpu-baseline --no-traffic --no-beyond-cache "$@" > $OUT/write.log 2>&1
grep '^{' $OUT/stats.log | tail -1 > $OUT/bench_line.json
echo "profile $TAG done"
