#!/bin/bash
# Counter passes for kernels A and B of one workload, one rocprofv3 run per pass (each pass stays inside
# the per-block counter slots: 8 SQ, 4 TCC, 4 TCP, 2 TA).  What is profiled: bench.py --pmc-child, i.e. the
# solver on the workload for six plain iterations.  Results: gpurun_out/pmc_<tag>/<pass>/...;
# tools/pmc_table.py <tag> prints them side by side and writes profiles/<tag>_pmc.json.
# usage: tools/pmc_passes.sh <tag> <workload>     (environment CG_* switches apply to the profiled run)
set -e
TAG=$1; WL=$2
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
pass() {
  name=$1; shift
  timeout -s KILL 120 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT/$name -- python3 $ROOT/bench.py --pmc-child --workload $WL > $OUT/$name.log 2>&1
}
pass sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
pass sq2 SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU SQ_INSTS_SALU
pass sq3 SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_WAIT_INST_LDS SQ_LEVEL_WAVES SQ_INSTS_SMEM SQ_BUSY_CU_CYCLES SQ_CYCLES
pass tcc1 TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum
pass tcc2 TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_STALL_sum TCC_TAG_STALL_sum
pass tcc3 TCC_REQ_sum TCC_READ_sum TCC_WRITE_sum TCC_STREAMING_REQ_sum
pass tcp1 TCP_PENDING_STALL_CYCLES_sum TCP_LFIFO_STALL_CYCLES_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum
pass fetch FETCH_SIZE
pass write WRITE_SIZE
echo "pmc passes $TAG done"
