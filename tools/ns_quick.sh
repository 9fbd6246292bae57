#!/bin/bash
# tools/ns_quick.sh [mask variants...] -- (GPU box) the NoiseSup iteration loop: parity of every kernel form + the
# full-size corpus, role periods of the given ablate/ variants, three bench lines of the in-tree library.
cd ${GRAFT_REPO_ROOT:-/root/repo}
python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_golden.py -q -x -k "ns_ or noisesup or etsi_denoise or golden or compceps or rfft or irm or afe" 2>&1 | tail -4
bash tools/ns_role_periods.sh "$@"
for i in 1 2 3; do python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-also 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('bench', round(d['ms_per_step'],3), 'ms', round(d['value']/1e6,1), 'M frames/s')"; done
