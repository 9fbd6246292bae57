#!/bin/bash
# tools/pmc_quick.sh -- (GPU box) vector / scalar / LDS instruction counts per launch of the NoiseSup kernels on
# configs[1] and on the configs[4] shard (counter-only passes)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
for cfg in "" "--corpus-utts 100000"; do
  rm -rf /tmp/pq
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_INSTS_VALU_FMA_F64 \
      --output-format csv -d /tmp/pq -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-also $cfg > /tmp/pq.log 2>&1 || { tail -3 /tmp/pq.log; continue; }
  echo "== bench $cfg"
  python3 $R/tools/prof_summary.py /tmp/pq /tmp/pq_sum.txt --delete-raw | grep -E "SQ_" | sed 's/sea::\([a-z0-9_]*\)(.*) /\1 /; s/dispatches=[0-9]* //; s/ min=.*//'
done
