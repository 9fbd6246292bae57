R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for form in ${FORMS:-pipe pre}; do
  export SEA_NS_KERNEL=$form
  rm -rf /tmp/pf_$form
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_BUSY_CYCLES \
      --output-format csv -d /tmp/pf_$form -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-also > /tmp/pf_$form.log 2>&1 || { tail -3 /tmp/pf_$form.log; continue; }
  echo "== $form"
  python3 $R/tools/prof_summary.py /tmp/pf_$form /tmp/pf_$form.txt --delete-raw | grep -E "SQ_" | sed 's/sea::\([a-z0-9_]*\)(.*) /\1 /; s/dispatches=[0-9]* //; s/ min=.*//'
  python3 $R/bench.py --steps 10 --no-cpu-baseline --no-also 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$form', round(d['ms_per_step'],3), 'ms', d['roofline']['kernel'])"
done
