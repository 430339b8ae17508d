# Regenerates round 4's headline artifacts under gpurun_out/ (copy the ones to keep into profiles/).
# usage (GPU box): bash profiles/scripts/refresh_artifacts_r04.sh TAG
TAG=${1:-r04_final}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; O=gpurun_out; mkdir -p $O; export TMPDIR=/tmp
# 1. the GPU suite
timeout 1500 python -m pytest tests -m gpu -q --maxfail=10 -p no:cacheprovider > $O/${TAG}_pytest_gpu.log 2>&1; tail -3 $O/${TAG}_pytest_gpu.log
T=16000000000      # the default workload's total target count
# 2. HBM traffic of the step's kernels on the default workload: FETCH_SIZE and WRITE_SIZE in passes of their own (kernel trace only)
( cd /tmp
  for pass in "d FETCH_SIZE" "e WRITE_SIZE"; do set -- $pass
    timeout 500 rocprofv3 --pmc $2 --kernel-trace --output-format csv -d $R/$O/pmc_${TAG}_$1 -- python $R/bench.py --steps 1 --warmup 1 --no-parity --no-legs > $R/$O/pmc_${TAG}_$1.log 2>&1; echo "pmc pass $1 rc=$?"
  done )
python profiles/scripts/pmc_summary.py $O/pmc_${TAG}_d $O/pmc_${TAG}_e > $O/${TAG}_pmc_counters.tsv 2> $O/${TAG}_pmc_summary.err
python profiles/scripts/make_pmc_traffic.py $O/pmc_${TAG} 10000000 150 $T 1 "profiles/${TAG}_pmc_counters.tsv (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of bench.py --steps 1 --warmup 1 --no-parity --no-legs)" diversity > $O/${TAG}_pmc_traffic_print.json 2> $O/${TAG}_pmc_traffic.err; cp profiles/pmc_traffic_diversity.json $O/${TAG}_pmc_traffic_diversity.json
find $O -name "*counter_collection.csv" -size +20M -delete
# 3. rocprofv3 kernel stats of the same workload
rm -rf $O/prof_ks && mkdir -p $O/prof_ks
( cd /tmp && timeout 500 rocprofv3 --kernel-trace --stats -d $R/$O/prof_ks -o ks -- python $R/bench.py --steps 5 --warmup 2 --no-parity --no-legs > $R/$O/prof_ks/bench.json 2> $R/$O/prof_ks/bench.log )
python profiles/scripts/rocpd_summary.py $(find $O/prof_ks -name "*.db" | head -1) > $O/${TAG}_rocprofv3_kernel_stats.txt 2>&1; head -14 $O/${TAG}_rocprofv3_kernel_stats.txt | cut -c1-150
find $O/prof_ks -name "*.db" -size +30M -delete
# 4. the headline line (after the counter passes: the line then carries `traffic` / `effective` of this very workload) (the driver's command: CPU baseline, parity samples, legs of the other configurations)
timeout 900 python bench.py --steps 20 --warmup 5 > $O/${TAG}_bench.json 2> $O/${TAG}_bench.log; echo "bench rc=$?"; grep "stage ms\|leg \|parity\|setup" $O/${TAG}_bench.log | cut -c1-250
# 5. continuity with round 3's headline: reads from 24 genomes, uniform candidate runs (no conserved segments)
timeout 400 python bench.py --species 24 --steps 10 --warmup 3 --no-legs --cpu-reads 200000 --no-cpu > $O/${TAG}_bench_24genomes.json 2> $O/${TAG}_bench_24genomes.log; grep "stage ms\|parity" $O/${TAG}_bench_24genomes.log | cut -c1-250
# 6. a database of 8 G targets from files, end to end; and the 204 M-target one of round 3
timeout 600 python profiles/scripts/e2e_big.py 8e9 60e6 64 4000000 > $O/${TAG}_e2e_8G_targets.txt 2>&1; grep "mtb_classify\|max-reads\|database" $O/${TAG}_e2e_8G_targets.txt | cut -c1-400
timeout 400 python profiles/scripts/e2e_big.py 2.04e8 60e6 64 2000000,4000000 > $O/${TAG}_e2e_204M_targets.txt 2>&1; grep "mtb_classify\|max-reads\|database" $O/${TAG}_e2e_204M_targets.txt | cut -c1-400
