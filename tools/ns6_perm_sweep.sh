#!/bin/bash
# tools/ns6_perm_sweep.sh <file of octal wave->role maps> -- (GPU box) configs[1] step of the six-wave form for each map (SEA_NS6_PERM)
cd ${GRAFT_REPO_ROOT:-/root/repo}
for p in $(cat $1); do
  SEA_NS6_PERM=$p SEA_NS_KERNEL=pipe6 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-also --no-configs4 --no-end-to-end 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$p', round(d['ms_per_step'],3), round(d['value']/1e6,1))"
done
