for so in "" ablate/libsea_rsA.so ablate/libsea_rsB.so; do
  if [ -n "$so" ]; then export SEA_MI355X_LIB=$PWD/$so; else unset SEA_MI355X_LIB; fi
  echo "== ${so:-baseline}"; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof/x -- python3 tools/bench_extra.py --utts 1024 --steps 2 --what resynth > /dev/null 2>&1; python tools/prof_summary.py /tmp/prof/x /tmp/x.txt --delete-raw | grep "resynth_.*n=" | cut -c1-90
done
