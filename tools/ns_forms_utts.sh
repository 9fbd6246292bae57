#!/bin/bash
# tools/ns_forms_utts.sh <utts...> -- (GPU box) bench.py step of the in-tree library for the four-wave, six-wave and table-in-LDS forms at several batch sizes
cd ${GRAFT_REPO_ROOT:-/root/repo}
for n in "$@"; do for k in pipe pipe6 big; do
  SEA_NS_KERNEL=$k python bench.py --utts $n --steps 10 --warmup 3 --no-cpu-baseline --no-also --no-configs4 --no-end-to-end 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print($n, '$k', round(d['ms_per_step'],3), 'ms', round(d['value']/1e6,1), 'M frames/s')"
done; done
