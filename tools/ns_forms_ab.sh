#!/bin/bash
# tools/ns_forms_ab.sh "<utts...>" <form:variant ...> -- (GPU box) bench.py step for pairs of (kernel form, ablate/libsea_<variant>.so), alternating, 3 rounds
cd ${GRAFT_REPO_ROOT:-/root/repo}
U="$1"; shift
for n in $U; do for r in 1 2 3; do for fv in "$@"; do
  k=${fv%%:*}; v=${fv##*:}
  SEA_MI355X_LIB=$PWD/ablate/libsea_$v.so SEA_NS_KERNEL=$k python bench.py --utts $n --steps 10 --warmup 3 --no-cpu-baseline --no-also --no-configs4 --no-end-to-end 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print($n, '$fv', round(d['ms_per_step'],3), 'ms', round(d['value']/1e6,1), 'M frames/s')"
done; done; done
