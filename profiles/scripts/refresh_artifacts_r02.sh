# Regenerates the round's headline artifacts under gpurun_out/ (copy the ones to keep into profiles/).
# usage (GPU box): bash profiles/scripts/refresh_artifacts_r02.sh TAG
TAG=${1:-r02}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; O=gpurun_out; mkdir -p $O
export TMPDIR=/tmp
python bench.py --steps 10 --warmup 3 > $O/${TAG}_bench.json 2> $O/${TAG}_bench.log; tail -2 $O/${TAG}_bench.log | cut -c1-400
rm -rf $O/prof_ks && mkdir -p $O/prof_ks
( cd /tmp && rocprofv3 --kernel-trace --stats -d $R/$O/prof_ks -o ks -- python $R/bench.py --steps 5 --warmup 2 --no-parity > $R/$O/prof_ks/bench.json 2> $R/$O/prof_ks/bench.log )
python profiles/scripts/rocpd_summary.py $(find $O/prof_ks -name "*.db" | head -1) > $O/${TAG}_rocprofv3_kernel_stats.txt 2>&1; head -14 $O/${TAG}_rocprofv3_kernel_stats.txt
bash profiles/scripts/pmc_sq.sh $TAG > /dev/null
python profiles/scripts/pmc_summary.py $O/pmc_${TAG}_a $O/pmc_${TAG}_b $O/pmc_${TAG}_c $O/pmc_${TAG}_d $O/pmc_${TAG}_e > $O/${TAG}_pmc_counters.tsv
T=$(python -c "import json; print(json.load(open('$O/${TAG}_bench.json'))['config']['targets'])")
python profiles/scripts/make_pmc_traffic.py $O/pmc_${TAG} 10000000 150 $T 1 "profiles/${TAG}_pmc_counters.tsv (rocprofv3 --pmc passes of bench.py --steps 1 --warmup 1 --no-parity)" > $O/${TAG}_pmc_traffic_print.json; cp profiles/pmc_traffic.json $O/${TAG}_pmc_traffic.json   # copy THIS one to profiles/pmc_traffic.json
python bench.py --cpu-reads 1000000 --steps 3 --warmup 3 --seq-mode 2 --reads 12500000 > $O/${TAG}_bench_paired.json 2> $O/${TAG}_bench_paired.log; grep "stage ms" $O/${TAG}_bench_paired.log
python bench.py --cpu-reads 20000 --steps 2 --warmup 2 --seq-mode 3 --reads 200000 --read-len 10000 > $O/${TAG}_bench_long.json 2> $O/${TAG}_bench_long.log; grep "stage ms" $O/${TAG}_bench_long.log; tail -3 $O/${TAG}_bench_long.log | cut -c1-300
find $O -name "*counter_collection.csv" -size +20M -delete
