# Round 4 artifacts, second take (the first one's gpurun_out exceeded the 64 MiB that travel back: nothing but its stdout came home).
# Same steps without the GPU suite, and only summaries are left under gpurun_out/.
TAG=${1:-r04_final}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; O=gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
S=/tmp/mtb_prof_scratch; rm -rf $S; mkdir -p $S
T=16000000000
( cd /tmp
  for pass in "d FETCH_SIZE" "e WRITE_SIZE"; do set -- $pass
    timeout 400 rocprofv3 --pmc $2 --kernel-trace --output-format csv -d $S/pmc_$1 -- python $R/bench.py --steps 1 --warmup 1 --no-parity --no-legs > $O/pmc_$1.log 2>&1; echo "pmc pass $1 rc=$?"
  done )
python profiles/scripts/pmc_summary.py $S/pmc_d $S/pmc_e > $O/${TAG}_pmc_counters.tsv 2> $O/pmc_summary.err
python profiles/scripts/make_pmc_traffic.py $S/pmc 10000000 150 $T 1 "profiles/${TAG}_pmc_counters.tsv (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of bench.py --steps 1 --warmup 1 --no-parity --no-legs)" diversity > $O/pmc_traffic_print.json 2> $O/pmc_traffic.err; cp profiles/pmc_traffic_diversity.json $O/pmc_traffic_diversity.json
rm -rf $S/pmc_d $S/pmc_e
( cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d $S/prof_ks -o ks -- python $R/bench.py --steps 5 --warmup 2 --no-parity --no-legs > $O/ks_bench.json 2> $O/ks_bench.log )
python profiles/scripts/rocpd_summary.py $(find $S/prof_ks -name "*.db" | head -1) > $O/${TAG}_rocprofv3_kernel_stats.txt 2>&1; head -8 $O/${TAG}_rocprofv3_kernel_stats.txt | cut -c1-150
rm -rf $S
timeout 600 python bench.py --steps 20 --warmup 5 > $O/${TAG}_bench.json 2> $O/${TAG}_bench.log; echo "bench rc=$?"; grep "stage ms\|leg \|parity\|setup" $O/${TAG}_bench.log | cut -c1-250
timeout 300 python bench.py --species 24 --steps 10 --warmup 3 --no-legs --cpu-reads 200000 --no-cpu > $O/${TAG}_bench_24genomes.json 2> $O/${TAG}_bench_24genomes.log; grep "stage ms\|parity" $O/${TAG}_bench_24genomes.log | cut -c1-250
du -sh $O
