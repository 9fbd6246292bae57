#!/bin/bash
# tools/profile_round.sh [tag] -- everything the profiles/ directory is made of, in one GPU-box call:
# bench line, extra-kernel bench lines, kernel-trace stats of both, HBM FETCH/WRITE passes (separate
# --pmc runs, no trace domains mixed in).  Summaries land in gpurun_out/<tag>_*; copy them to profiles/.
set -e
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
python bench.py > $O/${TAG}_bench_n1.json 2> $O/${TAG}_bench_n1.err
python tools/bench_extra.py --steps 5 > $O/${TAG}_bench_extra_1024.jsonl 2> /dev/null
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_ns -- python3 $R/bench.py --steps 10 --no-cpu-baseline > /tmp/p_ns.log 2>&1
python3 $R/tools/prof_summary.py /tmp/p_ns $O/${TAG}_ns_kernel_trace_stats.txt --delete-raw > /dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_ex -- python3 $R/tools/bench_extra.py --what resynth,ibm,subband,ceps,rfft --steps 4 > /tmp/p_ex.log 2>&1
python3 $R/tools/prof_summary.py /tmp/p_ex $O/${TAG}_extra_kernels_trace_stats.txt --delete-raw > /dev/null
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d /tmp/p_$c -- python3 $R/tools/bench_extra.py --what resynth,subband --steps 2 > /tmp/p_$c.log 2>&1
  python3 $R/tools/prof_summary.py /tmp/p_$c $O/${TAG}_resynth_pmc_$(echo $c | tr A-Z a-z | sed s/_size//).txt --delete-raw > /dev/null
  rocprofv3 --pmc $c --output-format csv -d /tmp/q_$c -- python3 $R/bench.py --steps 4 --no-cpu-baseline > /tmp/q_$c.log 2>&1
  python3 $R/tools/prof_summary.py /tmp/q_$c $O/${TAG}_ns_pmc_$(echo $c | tr A-Z a-z | sed s/_size//).txt --delete-raw > /dev/null
done
echo profile_round done
