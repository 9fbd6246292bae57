#!/bin/bash
# tools/gpu_check.sh [tag] -- one GPU-box call: the -m gpu tests, then the default bench line.
# A step that times out ends the call (no further GPU step after a hang).
TAG=${1:-r02}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $O/${TAG}_gpu_tests.log 2>&1
rc=$?
tail -15 $O/${TAG}_gpu_tests.log
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "gpu tests timed out: stopping"; exit $rc; fi
timeout -k 10 400 python bench.py > $O/${TAG}_bench_n1.json 2> $O/${TAG}_bench_n1.err
brc=$?
echo "bench rc=$brc"
cut -c1-1500 $O/${TAG}_bench_n1.json
tail -5 $O/${TAG}_bench_n1.err
exit $(( rc != 0 ? rc : brc ))
