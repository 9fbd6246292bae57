#!/bin/bash
# tools/ab_extra.sh <what> [rounds] -- tools/bench_extra.py --what <what> for the in-tree library and every ablate/*.so, alternating
W=${1:-rfft}; R=${2:-2}
for i in $(seq $R); do
  for so in "" ablate/*.so; do
    [ "$so" = "ablate/*.so" ] && continue
    if [ -n "$so" ]; then export SEA_MI355X_LIB=$PWD/$so; else unset SEA_MI355X_LIB; fi
    python tools/bench_extra.py --what $W --steps 10 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); r=d.get('roofline',{}); print('${so:-intree}', d['metric'][:40], round(r.get('avg_step_ms', d['ms_per_step']),4), 'ms', round(r.get('achieved',0),1), 'GB/s')"
  done
done
