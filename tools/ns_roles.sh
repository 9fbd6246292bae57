#!/bin/bash
# tools/ns_roles.sh <variant> -- (GPU box) role work / barrier-wait cycles per frame of ablate/libsea_<variant>.so
# (a -DSEA_NS_TIMING [-DSEA_NS_TIMING_NOCK] build), in shader clocks: independent of the box's clock
cd ${GRAFT_REPO_ROOT:-/root/repo}
for v in "$@"; do
SEA_MI355X_LIB=ablate/libsea_$v.so python tools/ns_timing.py 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('$v:', ' '.join(f\"{k} {d[k]['work_cyc_per_frame']}+{d[k]['wait_cyc_per_frame']}\" for k in ('F', 'B0', 'B1', 'S')), '| period', d['frame_period_ns'], 'ns at', d['shader_clock_MHz_during_the_launch'], 'MHz =', round(d['frame_period_ns'] * d['shader_clock_MHz_during_the_launch'] / 1000), 'clk | S ck', d['S_checkpoints_cyc_per_frame(prep,chains,energy,verify,store)'])"
done
