#!/bin/bash
# tools/ab_extra_v.sh <what> <variants...> -- (GPU box) tools/bench_extra.py --what <what> for ablate/libsea_<variant>.so, alternating, 2 rounds
cd ${GRAFT_REPO_ROOT:-/root/repo}
W=$1; shift
for r in 1 2; do for v in "$@"; do
  SEA_MI355X_LIB=$PWD/ablate/libsea_$v.so python tools/bench_extra.py --what $W --steps 8 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); r=d.get('roofline',{}); print('$v', d['metric'][:44], round(r.get('avg_step_ms', d['ms_per_step']),4), 'ms')"
done; done
