#!/bin/bash
# tools/ab_shard_forms.sh <form:variant ...> -- (GPU box) the configs[4] shard for pairs of (kernel form, ablate/libsea_<variant>.so), alternating, 2 rounds
cd ${GRAFT_REPO_ROOT:-/root/repo}
for r in 1 2; do for fv in "$@"; do
  k=${fv%%:*}; v=${fv##*:}
  SEA_NS_KERNEL=$k SEA_MI355X_LIB=$PWD/ablate/libsea_$v.so timeout -k 10 120 python bench.py --corpus-utts 100000 --steps 2 --warmup 1 --no-cpu-baseline --no-also --no-configs4 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$fv', round(d['ms_per_step'],3), 'ms', round(d['value']/1e6,1), 'M frames/s')"
done; done
