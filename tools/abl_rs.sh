#!/bin/bash
# timing-only: resynthesis per-tile period with pipeline roles switched off (ablate/libsea_rs_skip<mask>.so;
# bit 0 = R1, 1 = R2, 2 = H; outputs are wrong by construction).  Uses the equal-length probe.
for so in "" ablate/libsea_rs_skip*.so; do
  if [ -n "$so" ]; then export SEA_MI355X_LIB=$PWD/$so; else unset SEA_MI355X_LIB; fi
  python tools/rs_probe.py 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l)
        if d['n_utt'] in (256,1024): print('${so:-baseline}', d['n_utt'], 'resynth ns/tile (both passes)', round(d['resynth_ns_per_tile_both_passes']))
"
done
