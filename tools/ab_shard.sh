#!/bin/bash
# tools/ab_shard.sh [rounds] -- the configs[4] shard (12 500 utterances) for the in-tree library and every ablate/*.so, alternating
R=${1:-2}
for i in $(seq $R); do
  for so in "" ablate/*.so; do
    [ "$so" = "ablate/*.so" ] && continue
    if [ -n "$so" ]; then export SEA_MI355X_LIB=$PWD/$so; else unset SEA_MI355X_LIB; fi
    python bench.py --corpus-utts 100000 --steps 3 --warmup 1 --no-cpu-baseline --no-also 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('${so:-intree}', round(d['ms_per_step'],3), 'ms', round(d['value']/1e6,1), 'M frames/s', d['roofline']['kernel'])"
  done
done
