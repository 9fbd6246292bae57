#!/bin/bash
# tools/ns_role_periods.sh [libs...] -- lone frame period (ns per frame of one utterance, 128 equal utterances, four-wave
# form) of each ablate/libsea_<name>.so given: with -DSEA_ROLE_MASK variants (tools/build_variant.sh) this is each
# role's dependent-chain time by itself, without timers in the instruction stream.
cd ${GRAFT_REPO_ROOT:-/root/repo}
for so in "$@"; do
  echo -n "$so: "
  SEA_MI355X_LIB=ablate/libsea_$so.so SEA_NS_KERNEL=pipe python tools/ns_period.py 128 1024 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('lone', d['128']['ns_per_frame_of_one_utt'], 'ns; 1024:', d['1024']['ns_per_frame_of_one_utt'], 'ns', d['1024']['Mframes_s'], 'M/s')"
done
