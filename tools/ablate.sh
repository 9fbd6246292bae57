#!/bin/bash
# timing-only: bench every diagnostic variant in ablate/ (outputs are wrong by construction)
for so in "" ablate/*.so; do
  if [ -n "$so" ]; then export SEA_MI355X_LIB=$PWD/$so; else unset SEA_MI355X_LIB; fi
  python bench.py --steps 5 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('${so:-baseline}', round(d['ms_per_step'],3), 'ms')"
done
