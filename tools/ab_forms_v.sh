#!/bin/bash
# tools/ab_forms_v.sh <form> <variants...> -- (GPU box) configs[1] bench steps with SEA_NS_KERNEL=<form> for ablate/libsea_<variant>.so, alternating, 2 rounds
cd ${GRAFT_REPO_ROOT:-/root/repo}
F=$1; shift
for r in 1 2; do for v in "$@"; do
  SEA_NS_KERNEL=$F SEA_MI355X_LIB=$PWD/ablate/libsea_$v.so timeout -k 10 120 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-also --no-configs4 --no-end-to-end 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$F $v', round(d['ms_per_step'],4), 'ms', round(d['value']/1e6,1), 'M frames/s')"
done; done
