#!/bin/bash
# tools/profile_round.sh [tag] -- everything the profiles/ directory is made of, in one GPU-box call: the bench line
# (with its also-array), kernel-trace stats of the same command, HBM FETCH / WRITE counter passes (separate --pmc
# runs, no trace domains mixed in) -> profiles/pmc_traffic.json stamped with the kernel-source sha, SQ counter
# passes of the NoiseSup kernel.  Summaries land in gpurun_out/<tag>_*; copy them to profiles/.
TAG=${1:-r02}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
python bench.py > $O/${TAG}_bench_n1.json 2> $O/${TAG}_bench_n1.err || { tail -5 $O/${TAG}_bench_n1.err; exit 1; }
python tools/bench_extra.py --what subband,afe,host --steps 5 > $O/${TAG}_bench_extra_1024.jsonl 2> /dev/null
python tools/ns16k_time.py 1024 400 > $O/${TAG}_ns16k_time.txt 2> /dev/null
python tools/ns16k_time.py 4096 200 >> $O/${TAG}_ns16k_time.txt 2> /dev/null
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/p_tr
# every profiler pass prints a line first (a silent call is taken to be hung after 7 minutes) and runs under its own limit:
# a pass that hangs ends the script -- no further GPU step after a timeout
mkdir -p $O/${TAG}_logs
# the logs of the passes travel back too (profiles/r04_hang_record.md: round 3's stuck pass left nothing to read): whatever a
# pass has printed so far is copied next to the summaries when it ends, however it ends
keep_logs() { cp -f /tmp/p_*.log $O/${TAG}_logs/ 2> /dev/null; }
trap keep_logs EXIT
pass() { timeout -k 10 240 "$@" || { echo "profile_round: that pass failed or timed out: stopping" >&2; exit 1; }; }
echo "profile_round: kernel trace"; pass rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_tr -- python3 $R/bench.py --steps 10 --no-cpu-baseline --no-end-to-end > /tmp/p_tr.log 2>&1
python3 $R/tools/prof_summary.py /tmp/p_tr $O/${TAG}_kernel_trace_stats.txt --delete-raw > /dev/null
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/p_$c
  echo "profile_round: pmc $c"; pass rocprofv3 --pmc $c --output-format csv -d /tmp/p_$c -- python3 $R/bench.py --steps 3 --warmup 1 --also-steps 2 --no-cpu-baseline --no-end-to-end --no-configs4 > /tmp/p_$c.log 2>&1
  python3 $R/tools/prof_summary.py /tmp/p_$c $O/${TAG}_pmc_$(echo $c | tr A-Z a-z | sed s/_size//).txt --delete-raw > /dev/null
done
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_ANY" \
           "SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_SCA"; do
  i=$((i+1))
  rm -rf /tmp/p_sq$i
  echo "profile_round: pmc SQ set $i"; pass rocprofv3 --pmc $set --output-format csv -d /tmp/p_sq$i -- python3 $R/bench.py --steps 3 --warmup 1 --also-steps 2 --no-cpu-baseline --no-end-to-end --no-configs4 > /tmp/p_sq$i.log 2>&1
  python3 $R/tools/prof_summary.py /tmp/p_sq$i $O/${TAG}_pmc_sq$i.txt --delete-raw > /dev/null
done
# the configs[4] kernel form (sea::ns_denoise_pipe_big_kernel: the form all eight GPUs run) gets counter passes of its own on
# the workload it is quoted on: LPT shard 0 of 8 of the 100000-utterance corpus, 12500 utterances, one launch per step
BIG="--corpus-utts 100000 --steps 2 --warmup 1 --no-also --no-cpu-baseline"
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/p_big_$c
  echo "profile_round: pmc $c, configs[4] form"; pass rocprofv3 --pmc $c --output-format csv -d /tmp/p_big_$c -- python3 $R/bench.py $BIG > /tmp/p_big_$c.log 2>&1
  python3 $R/tools/prof_summary.py /tmp/p_big_$c $O/${TAG}_pmc_big_$(echo $c | tr A-Z a-z | sed s/_size//).txt --delete-raw > /dev/null
done
rm -rf /tmp/p_big_sq1 /tmp/p_big_mix1
echo "profile_round: pmc SQ set 1, configs[4] form"; pass rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_ANY \
    --output-format csv -d /tmp/p_big_sq1 -- python3 $R/bench.py $BIG > /tmp/p_big_sq1.log 2>&1
python3 $R/tools/prof_summary.py /tmp/p_big_sq1 $O/${TAG}_pmc_big_sq1.txt --delete-raw > /dev/null
rm -rf /tmp/p_big_sq2
echo "profile_round: pmc SQ set 2 (LDS), configs[4] form"; pass rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_SCA \
    --output-format csv -d /tmp/p_big_sq2 -- python3 $R/bench.py $BIG > /tmp/p_big_sq2.log 2>&1
python3 $R/tools/prof_summary.py /tmp/p_big_sq2 $O/${TAG}_pmc_big_sq2.txt --delete-raw > /dev/null
echo "profile_round: pmc instruction mix, configs[4] form"; pass rocprofv3 --pmc SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 \
    --output-format csv -d /tmp/p_big_mix1 -- python3 $R/bench.py $BIG > /tmp/p_big_mix1.log 2>&1
python3 $R/tools/prof_summary.py /tmp/p_big_mix1 $O/${TAG}_pmc_big_mix1.txt --delete-raw > /dev/null
# dynamic instruction mix of the headline kernel (prices its vector-issue time: roofline.valu_issue_frac)
rm -rf /tmp/p_mix1
echo "profile_round: pmc instruction mix"; pass rocprofv3 --pmc SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 \
    --output-format csv -d /tmp/p_mix1 -- python3 $R/bench.py --steps 3 --warmup 1 --no-also --no-cpu-baseline > /tmp/p_mix1.log 2>&1
python3 $R/tools/prof_summary.py /tmp/p_mix1 $O/${TAG}_pmc_mix1.txt --delete-raw > /dev/null
python3 $R/tools/make_pmc_traffic.py $O/${TAG}_pmc_fetch.txt $O/${TAG}_pmc_write.txt $O/pmc_traffic.json $O/${TAG}_pmc_sq1.txt $O/${TAG}_pmc_mix1.txt \
    --big $O/${TAG}_pmc_big_fetch.txt $O/${TAG}_pmc_big_write.txt $O/${TAG}_pmc_big_sq1.txt $O/${TAG}_pmc_big_mix1.txt && cp $O/pmc_traffic.json $R/profiles/pmc_traffic.json
cd $R
python bench.py --no-cpu-baseline > $O/${TAG}_bench_n1_with_traffic.json 2> /dev/null   # same tree, traffic from the fresh stamp
echo "profile_round done -- back home copy gpurun_out/${TAG}_* AND gpurun_out/pmc_traffic.json into profiles/ (the stamp bench.py checks)"
