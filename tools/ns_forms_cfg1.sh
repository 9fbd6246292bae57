#!/bin/bash
# tools/ns_forms_cfg1.sh <form ...> -- (GPU box) bench.py steps on configs[1] for the given SEA_NS_KERNEL forms, alternating, 2 rounds
cd ${GRAFT_REPO_ROOT:-/root/repo}
for r in 1 2; do for k in "$@"; do
  SEA_NS_KERNEL=$k timeout -k 10 120 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-also --no-configs4 --no-end-to-end 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$k', round(d['ms_per_step'],4), 'ms', round(d['value']/1e6,1), 'M frames/s')"
done; done
