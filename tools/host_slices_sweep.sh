#!/bin/bash
# tools/host_slices_sweep.sh -- (GPU box) sea_denoise_utterances on the bench corpus: time slices (SEA_HOST_SLICES) against
# the chunk pipeline (SEA_HOST_MODE=chunks), alternating
cd ${GRAFT_REPO_ROOT:-/root/repo}
one() { python tools/bench_extra.py --what host --steps 10 2>/dev/null | grep HOST-buffer | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],2),'ms', round(d['value']/1e6,1),'M frames/s', d['config'].get('ms_per_call_sorted'))"; }
for r in 1 2; do
  echo -n "chunks: "; SEA_HOST_MODE=chunks one
  for k in ${SLICES:-2 3 4 6 8 12 16}; do echo -n "slices=$k: "; SEA_HOST_SLICES=$k one; done
done
