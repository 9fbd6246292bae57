#!/bin/bash
# tools/pmc_ns.sh -- SQ counter passes over the NoiseSup kernel (run on the GPU box).
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_ANY" \
           "SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_SCA"; do
  i=$((i+1))
  rm -rf /tmp/pmcn_$i
  rocprofv3 --pmc $set --output-format csv -d /tmp/pmcn_$i -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > /tmp/pmcn_$i.log 2>&1 || { tail -5 /tmp/pmcn_$i.log; echo "pass $i failed"; continue; }
  python3 $R/tools/prof_summary.py /tmp/pmcn_$i $R/gpurun_out/${TAG:-r01_ns_pipe}_pmc_sq$i.txt --delete-raw > /dev/null
done
