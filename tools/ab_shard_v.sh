#!/bin/bash
# tools/ab_shard_v.sh <variants...> -- (GPU box) the configs[4] shard (12 500 utterances, one launch per step) for ablate/libsea_<variant>.so, alternating, 2 rounds
cd ${GRAFT_REPO_ROOT:-/root/repo}
for r in 1 2; do for v in "$@"; do
  SEA_MI355X_LIB=$PWD/ablate/libsea_$v.so python bench.py --corpus-utts 100000 --steps 3 --warmup 1 --no-cpu-baseline --no-also --no-configs4 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$v', round(d['ms_per_step'],3), 'ms', round(d['value']/1e6,1), 'M frames/s', d['roofline']['kernel'])"
done; done
