#!/bin/bash
# tools/ab_cfg1_v.sh <variants...> -- (GPU box) bench.py steps on configs[1] for ablate/libsea_<variant>.so, alternating, 3 rounds
cd ${GRAFT_REPO_ROOT:-/root/repo}
for r in 1 2 3; do for v in "$@"; do
  SEA_MI355X_LIB=$PWD/ablate/libsea_$v.so timeout -k 10 120 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-also --no-configs4 --no-end-to-end 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$v', round(d['ms_per_step'],4), 'ms', round(d['value']/1e6,1), 'M frames/s')"
done; done
