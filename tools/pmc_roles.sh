#!/bin/bash
# tools/pmc_roles.sh -- dynamic instruction counts of the NoiseSup pipeline BY ROLE (run on the GPU box).
# For the in-tree library and every ablate/libsea_*.so (timing-only variants built with -DSEA_ROLE_MASK=...,
# tools/build_variant.sh) one SQ counter pass over bench.py; the difference to the full build is the role's share.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
for so in "" $R/ablate/libsea_*.so; do
  name=base
  if [ -n "$so" ]; then [ -f "$so" ] || continue; export SEA_MI355X_LIB=$so; name=$(basename $so .so | sed s/libsea_//); else unset SEA_MI355X_LIB; fi
  rm -rf /tmp/pr_$name
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_BUSY_CYCLES \
      --output-format csv -d /tmp/pr_$name -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-also > /tmp/pr_$name.log 2>&1 || { tail -3 /tmp/pr_$name.log; continue; }
  echo "== $name"
  python3 $R/tools/prof_summary.py /tmp/pr_$name $O/${TAG:-r02}_roles_$name.txt --delete-raw | grep -E "SQ_INSTS|SQ_ACTIVE|SQ_WAVE_CYCLES" | sed 's/sea::\([a-z0-9_]*\)(.*) /\1 /; s/dispatches=[0-9]* //; s/ min=.*//'
done
