#!/bin/bash
# tools/pmc_mix.sh -- dynamic VALU instruction mix and instruction-cache counters of the bench kernels (GPU box).
# Three counter-only passes over bench.py; summaries -> gpurun_out/${TAG:-r02}_pmc_mix{1,2,3}.txt
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
pass() {
  n=$1; shift
  rm -rf /tmp/pm_$n
  rocprofv3 --pmc "$@" --output-format csv -d /tmp/pm_$n -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline ${BENCH_FLAGS:---no-also} > /tmp/pm_$n.log 2>&1 || { tail -5 /tmp/pm_$n.log; return 1; }
  python3 $R/tools/prof_summary.py /tmp/pm_$n $O/${TAG:-r02}_pmc_mix$n.txt --delete-raw | grep -E "SQ" | sed 's/sea::\([a-z0-9_]*\)(.*) /\1 /; s/dispatches=[0-9]* //; s/ min=.*//'
}
pass 1 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 &&
pass 2 SQ_INSTS_VALU SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_INST_CYCLES_VALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES &&
pass 3 SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_INSTS_SMEM SQ_INSTS_SALU SQ_INST_CYCLES_SALU SQ_WAVE_CYCLES
