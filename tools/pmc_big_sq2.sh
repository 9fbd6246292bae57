cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
rm -rf /tmp/p_big_sq2
echo "pmc SQ set 2, configs[4] form"
timeout -k 10 300 rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_SCA --output-format csv -d /tmp/p_big_sq2 -- python3 $R/bench.py --corpus-utts 100000 --steps 2 --warmup 1 --no-also --no-cpu-baseline > /tmp/p_big_sq2.log 2>&1 || { tail -5 /tmp/p_big_sq2.log; exit 1; }
python3 $R/tools/prof_summary.py /tmp/p_big_sq2 $R/gpurun_out/r04_pmc_big_sq2.txt --delete-raw > /dev/null
cat $R/gpurun_out/r04_pmc_big_sq2.txt
