# Round 4 artifacts, third take: the counter passes and the kernel stats (the second take redirected them to a relative path from /tmp),
# the driver tests on the last driver change, the end-to-end runs.  Only summaries stay under gpurun_out/.
TAG=${1:-r04_final}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; O=$R/gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
S=/tmp/mtb_prof_scratch; rm -rf $S; mkdir -p $S
T=16000000000
( cd /tmp
  for pass in "d FETCH_SIZE" "e WRITE_SIZE"; do set -- $pass
    timeout 300 rocprofv3 --pmc $2 --kernel-trace --output-format csv -d $S/pmc_$1 -- python $R/bench.py --steps 1 --warmup 1 --no-parity --no-legs > $O/pmc_$1.log 2>&1; echo "pmc pass $1 rc=$?"
  done )
python profiles/scripts/pmc_summary.py $S/pmc_d $S/pmc_e > $O/${TAG}_pmc_counters.tsv 2> $O/pmc_summary.err
python profiles/scripts/make_pmc_traffic.py $S/pmc 10000000 150 $T 1 "profiles/${TAG}_pmc_counters.tsv (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of bench.py --steps 1 --warmup 1 --no-parity --no-legs)" diversity > $O/pmc_traffic_print.json 2> $O/pmc_traffic.err; cp profiles/pmc_traffic_diversity.json $O/pmc_traffic_diversity.json
rm -rf $S/pmc_d $S/pmc_e
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $S/prof_ks -o ks -- python $R/bench.py --steps 5 --warmup 2 --no-parity --no-legs > $O/ks_bench.json 2> $O/ks_bench.log )
python profiles/scripts/rocpd_summary.py $(find $S/prof_ks -name "*.db" | head -1) > $O/${TAG}_rocprofv3_kernel_stats.txt 2>&1; head -8 $O/${TAG}_rocprofv3_kernel_stats.txt | cut -c1-150
rm -rf $S
timeout 240 python -m pytest tests/test_gpu_driver.py tests/test_gpu_parity.py -m gpu -q -p no:cacheprovider -k "driver or chunk_by_chunk" > $O/pytest_driver.log 2>&1; tail -3 $O/pytest_driver.log
timeout 200 python profiles/scripts/e2e_big.py 2.04e8 60e6 64 2000000,4000000 > $O/${TAG}_e2e_204M_targets.txt 2>&1; grep "mtb_classify: 6\|max-reads" $O/${TAG}_e2e_204M_targets.txt | cut -c1-330
timeout 260 python profiles/scripts/e2e_big.py 8e9 60e6 64 4000000 > $O/${TAG}_e2e_8G_targets.txt 2>&1; grep "mtb_classify\|max-reads\|database" $O/${TAG}_e2e_8G_targets.txt | cut -c1-330
du -sh $O
