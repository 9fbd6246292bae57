#!/bin/bash
# tools/ns_ab4.sh <variants...> -- (GPU box) alternating bench lines on the configs[4] shard (12 500 utterances)
cd ${GRAFT_REPO_ROOT:-/root/repo}
for r in 1 2; do for v in "$@"; do
  SEA_MI355X_LIB=ablate/libsea_$v.so python bench.py --corpus-utts 100000 --steps 5 --warmup 2 --no-cpu-baseline --no-also 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$v', round(d['ms_per_step'],3), 'ms', round(d['value']/1e6,1), 'M frames/s')"
done; done
