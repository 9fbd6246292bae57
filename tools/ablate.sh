#!/bin/bash
# tools/ablate.sh [rounds] -- bench the in-tree library and every variant in ablate/, ALTERNATING, `rounds` times
# (default 3).  A/B differences below ~3 % only show with alternation and >= 20 steps per run: the first runs on a
# fresh box are slower, and 5-step runs scatter by +-2 %.  Variant outputs may be wrong by construction.
R=${1:-3}
for i in $(seq $R); do
  for so in "" ablate/*.so; do
    [ "$so" = "ablate/*.so" ] && continue
    if [ -n "$so" ]; then export SEA_MI355X_LIB=$PWD/$so; else unset SEA_MI355X_LIB; fi
    python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-also 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('${so:-baseline}', round(d['ms_per_step'],3), 'ms')"
  done
done
