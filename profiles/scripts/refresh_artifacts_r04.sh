# Regenerates round 4's artifacts (what profiles/r04_final_* hold).  Only SUMMARIES are left under gpurun_out/<TAG>/ -- the counter CSVs
# and the rocprofv3 databases live in /tmp and are deleted: gpurun brings back 64 MiB at most (the first take of this script lost
# everything but its stdout to that limit).  usage (GPU box): bash profiles/scripts/refresh_artifacts_r04.sh [TAG] [suite]
TAG=${1:-r04_final}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; O=$R/gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
S=/tmp/mtb_prof_scratch; rm -rf $S; mkdir -p $S
T=16000000000      # the default workload's total target count
# 0. (optional) the GPU suite
if [ "$2" = "suite" ]; then timeout 1500 python -m pytest tests -m gpu -q --maxfail=10 -p no:cacheprovider > $O/${TAG}_pytest_gpu.log 2>&1; tail -3 $O/${TAG}_pytest_gpu.log; fi
# 1. HBM traffic of the step's kernels on the default workload: FETCH_SIZE and WRITE_SIZE in passes of their own (kernel trace only)
( cd /tmp
  for pass in "d FETCH_SIZE" "e WRITE_SIZE"; do set -- $pass
    timeout 300 rocprofv3 --pmc $2 --kernel-trace --output-format csv -d $S/pmc_$1 -- python $R/bench.py --steps 1 --warmup 1 --no-parity --no-legs > $O/pmc_$1.log 2>&1; echo "pmc pass $1 rc=$?"
  done )
python profiles/scripts/pmc_summary.py $S/pmc_d $S/pmc_e > $O/${TAG}_pmc_counters.tsv 2> $O/pmc_summary.err
python profiles/scripts/make_pmc_traffic.py $S/pmc 10000000 150 $T 1 "profiles/${TAG}_pmc_counters.tsv (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of bench.py --steps 1 --warmup 1 --no-parity --no-legs)" diversity > $O/pmc_traffic_print.json 2> $O/pmc_traffic.err; cp profiles/pmc_traffic_diversity.json $O/pmc_traffic_diversity.json
rm -rf $S/pmc_d $S/pmc_e
# 2. rocprofv3 kernel stats of the same workload
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $S/prof_ks -o ks -- python $R/bench.py --steps 5 --warmup 2 --no-parity --no-legs > $O/ks_bench.json 2> $O/ks_bench.log )
python profiles/scripts/rocpd_summary.py $(find $S/prof_ks -name "*.db" | head -1) > $O/${TAG}_rocprofv3_kernel_stats.txt 2>&1; head -8 $O/${TAG}_rocprofv3_kernel_stats.txt | cut -c1-150
rm -rf $S
# 3. the headline line (the driver's command; after the counter passes, so that it carries `traffic` / `effective` of this workload)
timeout 600 python bench.py --steps 20 --warmup 5 > $O/${TAG}_bench.json 2> $O/${TAG}_bench.log; echo "bench rc=$?"; grep "stage ms\|leg \|parity\|setup" $O/${TAG}_bench.log | cut -c1-250
# 4. continuity with round 3's headline: reads from 24 genomes, uniform candidate runs; and 2400 genomes with uniform runs
timeout 300 python bench.py --species 24 --steps 10 --warmup 3 --no-legs --cpu-reads 200000 --no-cpu > $O/${TAG}_bench_24genomes.json 2> $O/${TAG}_bench_24genomes.log; grep "stage ms\|parity" $O/${TAG}_bench_24genomes.log | cut -c1-250
timeout 400 python bench.py --steps 5 --warmup 2 --no-conserved --no-legs --no-cpu --cpu-reads 100000 > $O/${TAG}_bench_uniform.json 2> $O/${TAG}_bench_uniform.log; grep "stage ms\|parity" $O/${TAG}_bench_uniform.log | cut -c1-250
# 5. databases from files, end to end with the stand-alone driver: 204 M targets (round 3's) and 8 G targets
timeout 200 python profiles/scripts/e2e_big.py 2.04e8 60e6 64 2000000,4000000 > $O/${TAG}_e2e_204M_targets.txt 2>&1; grep "mtb_classify: 6\|max-reads" $O/${TAG}_e2e_204M_targets.txt | cut -c1-330
timeout 300 python profiles/scripts/e2e_big.py 8e9 60e6 64 4000000 > $O/${TAG}_e2e_8G_targets.txt 2>&1; grep "mtb_classify\|max-reads\|database" $O/${TAG}_e2e_8G_targets.txt | cut -c1-330
du -sh $O
