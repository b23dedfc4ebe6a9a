#!/bin/bash
# All rocprofv3 profiles of a round in one go (run on the GPU box): tools/profile_bench.sh per workload, then a
# kernel-trace/stats pass of the mixed-precision solve.  usage: tools/profile_round.sh r02
set -e
R=$1
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $ROOT
tools/profile_bench.sh ${R}_2d8192 --workload 2d-8192
tools/profile_bench.sh ${R}_2d16384x8192 --workload 16384x8192
tools/profile_bench.sh ${R}_3d1024x128 --workload 3d-1024x128
tools/profile_bench.sh ${R}_3d512 --workload 3d-512
tools/profile_bench.sh ${R}_4d128morton --workload 4d-128-morton
tools/profile_bench.sh ${R}_4d128lex --workload 128x128x128x128
CG_MORTON_RELAYOUT=0 tools/profile_bench.sh ${R}_4d128morton_native --workload 4d-128-morton
OUT=$ROOT/gpurun_out/prof_${R}_mixed3d512
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $ROOT/bench.py --workload mixed-3d-512 --repeats 1 --warmup 0 > $OUT/stats.log 2>&1
grep '^{' $OUT/stats.log | tail -1 > $OUT/bench_line.json
echo "round profiles done"
