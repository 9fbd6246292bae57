#!/bin/bash
# tools/cc_ab.sh <variants...> -- (GPU box) CompCeps / AFE feature chain timings of ablate/libsea_<variant>.so
cd ${GRAFT_REPO_ROOT:-/root/repo}
for r in 1 2; do for v in "$@"; do
  SEA_MI355X_LIB=ablate/libsea_$v.so python tools/bench_extra.py --what ceps,afe --steps 10 2>/dev/null | python -c "
import sys,json
out=[]
for l in sys.stdin:
    d=json.loads(l); out.append(d['metric'][:14]+' '+str(round(d['ms_per_step'],3))+(' feat '+str(round(d['config'].get('waveproc_compceps_postproc_vad_ms',0),3)) if 'AFE' in d['metric'] else ''))
print('$v', ' | '.join(out))"
done; done
