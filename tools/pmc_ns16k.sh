#!/bin/bash
# tools/pmc_ns16k.sh [streams frames] -- (GPU box) SQ counters of the 16 k-native NoiseSup kernel on tools/ns16k_time.py's
# workload: instruction counts, LDS bank conflicts against LDS-active cycles (two counter-only passes)
R=${GRAFT_REPO_ROOT:-/root/repo}
B=${1:-1024}; NF=${2:-400}
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_ANY" \
           "SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_SCA"; do
  i=$((i+1))
  rm -rf /tmp/p16_$i
  echo "pmc_ns16k: SQ set $i"
  timeout -k 10 200 rocprofv3 --pmc $set --output-format csv -d /tmp/p16_$i -- python3 $R/tools/ns16k_time.py $B $NF > /tmp/p16_$i.log 2>&1 || { tail -5 /tmp/p16_$i.log; echo "pass $i failed"; exit 1; }
  python3 $R/tools/prof_summary.py /tmp/p16_$i /tmp/p16_sum$i.txt --delete-raw | grep -E "SQ_" | sed 's/sea::\([a-z0-9_]*\)(.*) /\1 /; s/dispatches=[0-9]* //; s/ min=.*//'
done
