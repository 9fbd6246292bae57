#!/bin/bash
# tools/ns_forms_ab.sh [forms...] -- (GPU box) alternating configs[1] bench lines of the in-tree library per kernel form
cd ${GRAFT_REPO_ROOT:-/root/repo}
for r in 1 2 3; do for f in "$@"; do
  SEA_NS_KERNEL=$f python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-also 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$f', round(d['ms_per_step'],3), 'ms', round(d['value']/1e6,1), 'M frames/s')"
done; done
