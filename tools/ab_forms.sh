#!/bin/bash
# tools/ab_forms.sh [rounds] [forms...] -- bench the kernel forms of the in-tree library, alternating
R=${1:-2}; shift
FORMS=${@:-pipe pre}
for i in $(seq $R); do
  for f in $FORMS; do
    SEA_NS_KERNEL=$f python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-also 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$f', round(d['ms_per_step'],3), 'ms', round(d['value']/1e6,1), 'M frames/s')"
  done
done
