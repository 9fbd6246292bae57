#!/bin/bash
# tools/host_sweep.sh -- (GPU box) the host-buffer NoiseSup pipeline against chunk size, issue order and output path
cd ${GRAFT_REPO_ROOT:-/root/repo}
python -m pytest tests/test_gpu_parity.py -q -x -k "pipeline_any_chunking or chunked_by_scratch or concurrent_host" > gpurun_out/host_tests.log 2>&1 || { tail -30 gpurun_out/host_tests.log; exit 1; }
tail -3 gpurun_out/host_tests.log
for zc in o n; do for ord in mix desc asc; do for mb in 8 16 24 32 48; do
    echo -n "zerocopy=$zc order=$ord chunk_mb=$mb: "
    SEA_HOST_ZEROCOPY=$zc SEA_HOST_ORDER=$ord SEA_HOST_CHUNK_MB=$mb python tools/bench_extra.py --what host --steps 10 2>/dev/null | grep HOST-buffer | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],2),'ms', round(d['value']/1e6,1),'M frames/s')"
done; done; done
