#!/bin/bash
# tools/pmc_resynth.sh -- SQ counter passes over the resynthesis kernels (run on the GPU box).
# Each pass is its own rocprofv3 run (counters only, no trace domains), condensed into gpurun_out/.
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_ANY" \
           "SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_INSTS_SMEM" \
           "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_VALU_TRANS"; do
  i=$((i+1))
  rm -rf /tmp/pmc_$i
  rocprofv3 --pmc $set --output-format csv -d /tmp/pmc_$i -- python3 $R/tools/bench_extra.py --what ${WHAT:-resynth} --steps 2 > /tmp/pmc_$i.log 2>&1 || { tail -5 /tmp/pmc_$i.log; echo "pass $i failed"; continue; }
  python3 $R/tools/prof_summary.py /tmp/pmc_$i $R/gpurun_out/${TAG:-r01_resynth}_pmc_sq$i.txt --delete-raw
done
