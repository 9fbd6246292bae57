#!/bin/bash
# tools/pmc_pair.sh [form ...] -- (GPU box) vector / scalar / LDS instruction counts and wait cycles per launch of the NoiseSup
# kernel forms (default: big pair) on the configs[4] shard (12 500 utterances), one counter-only pass per form
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for f in ${@:-big pair}; do
  rm -rf /tmp/pq
  echo "== SEA_NS_KERNEL=$f"
  SEA_NS_KERNEL=$f timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_WAIT_INST_LDS \
      --output-format csv -d /tmp/pq -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-also --corpus-utts 100000 > /tmp/pq.log 2>&1 || { tail -3 /tmp/pq.log; exit 1; }
  python3 $R/tools/prof_summary.py /tmp/pq /tmp/pq_sum.txt --delete-raw | grep -E "SQ_" | sed 's/sea::\([a-z0-9_]*\)(.*) /\1 /; s/dispatches=[0-9]* //; s/ min=.*//'
done
