#!/bin/bash
# Collects the profiles of one bench configuration on the GPU box:
#   [PASSES="bench stats 1 2 3 4 5 6"] profiles/collect.sh <tag> [bench.py flags...]
# writes gpurun_out/<tag>/{bench.json, stats/, pmc1..6/}.  Each counter pass is a run of its own.  The profiled
# runs render one frame at a time (--frames-in-flight 1), like the frames bench.py times its kernels on, so that
# kernel durations are not stretched by another frame's kernels sharing the GPU.
set -e -o pipefail
R=$(cd "$(dirname "$0")/.." && pwd)
TAG=$1; shift
O=$R/gpurun_out/$TAG
mkdir -p "$O"
PASSES=${PASSES:-"bench stats 1 2 3 4"}
want() { [[ " $PASSES " == *" $1 "* ]]; }
if want bench; then
  timeout -k 10 300 python3 "$R/bench.py" --steps 10 --warmup 3 "$@" > "$O/bench.json" 2> "$O/bench.err"
fi
cd /tmp && export TMPDIR=/tmp
if want stats; then
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/stats" -- python3 "$R/bench.py" --steps 10 --warmup 3 --no-cpu-baseline --frames-in-flight 1 "$@" > "$O/stats.log" 2>&1
fi
pass() {
  local n=$1; shift
  want "$n" || return 0
  timeout -s KILL 120 rocprofv3 --pmc "$@" --output-format csv -d "$O/pmc$n" -- python3 "$R/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --frames-in-flight 1 "${BENCH_ARGS[@]}" > "$O/pmc$n.log" 2>&1
}
BENCH_ARGS=("$@")
pass 1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU
pass 2 FETCH_SIZE
pass 3 WRITE_SIZE
pass 4 SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_INSTS_LDS
pass 5 SQC_DCACHE_HITS SQC_DCACHE_MISSES SQC_DCACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64
pass 6 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_INST_CYCLES_SMEM SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_INSTS_BRANCH
echo collected "$O"
