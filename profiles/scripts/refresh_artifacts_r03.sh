# Regenerates round 3's headline artifacts under gpurun_out/ (copy the ones to keep into profiles/).
# usage (GPU box): bash profiles/scripts/refresh_artifacts_r03.sh TAG
TAG=${1:-r03_final}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; O=gpurun_out; mkdir -p $O
export TMPDIR=/tmp
# 1. the GPU suite
timeout 1500 python -m pytest tests -m gpu -x -q > $O/${TAG}_pytest_gpu.log 2>&1; tail -3 $O/${TAG}_pytest_gpu.log
# 2. HBM traffic of the step's kernels: FETCH_SIZE and WRITE_SIZE in passes of their own (kernel trace only)
( cd /tmp
  for pass in "d FETCH_SIZE" "e WRITE_SIZE"; do set -- $pass
    timeout 600 rocprofv3 --pmc $2 --kernel-trace --output-format csv -d $R/$O/pmc_${TAG}_$1 -- python $R/bench.py --steps 1 --warmup 1 --no-parity > $R/$O/pmc_${TAG}_$1.log 2>&1; echo "pmc pass $1 rc=$?"
  done )
python profiles/scripts/pmc_summary.py $O/pmc_${TAG}_d $O/pmc_${TAG}_e > $O/${TAG}_pmc_counters.tsv 2> $O/${TAG}_pmc_summary.err
T=16024359718
python profiles/scripts/make_pmc_traffic.py $O/pmc_${TAG} 10000000 150 $T 1 "profiles/${TAG}_pmc_counters.tsv (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of bench.py --steps 1 --warmup 1 --no-parity)" > $O/${TAG}_pmc_traffic_print.json 2> $O/${TAG}_pmc_traffic.err; cp profiles/pmc_traffic.json $O/${TAG}_pmc_traffic.json
find $O -name "*counter_collection.csv" -size +20M -delete
# 3. the headline line (default command, CPU baseline and both parity checks included)
timeout 900 python bench.py --steps 10 --warmup 3 > $O/${TAG}_bench.json 2> $O/${TAG}_bench.log; grep "stage ms" $O/${TAG}_bench.log; tail -c 600 $O/${TAG}_bench.json
# 4. rocprofv3 kernel stats of the same command
rm -rf $O/prof_ks && mkdir -p $O/prof_ks
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $R/$O/prof_ks -o ks -- python $R/bench.py --steps 5 --warmup 2 --no-parity > $R/$O/prof_ks/bench.json 2> $R/$O/prof_ks/bench.log )
python profiles/scripts/rocpd_summary.py $(find $O/prof_ks -name "*.db" | head -1) > $O/${TAG}_rocprofv3_kernel_stats.txt 2>&1; head -12 $O/${TAG}_rocprofv3_kernel_stats.txt | cut -c1-150
find $O/prof_ks -name "*.db" -size +30M -delete
# 5. the other single-GPU configurations: bench line + kernel stats
timeout 600 python bench.py --cpu-reads 1000000 --cpu-targets 16e6 --steps 3 --warmup 2 --seq-mode 2 --reads 12500000 > $O/${TAG}_bench_paired.json 2> $O/${TAG}_bench_paired.log; grep "stage ms" $O/${TAG}_bench_paired.log
timeout 600 python bench.py --cpu-reads 20000 --cpu-targets 16e6 --steps 2 --warmup 2 --seq-mode 3 --reads 200000 --read-len 10000 > $O/${TAG}_bench_long.json 2> $O/${TAG}_bench_long.log; grep "stage ms" $O/${TAG}_bench_long.log
for cfg in "paired --seq-mode 2 --reads 12500000" "long --seq-mode 3 --reads 200000 --read-len 10000"; do
  set -- $cfg; name=$1; shift
  rm -rf $O/prof_$name && mkdir -p $O/prof_$name
  ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $R/$O/prof_$name -o ks -- python $R/bench.py --steps 3 --warmup 2 --no-parity "$@" > $R/$O/prof_$name/bench.json 2> $R/$O/prof_$name/bench.log )
  python profiles/scripts/rocpd_summary.py $(find $O/prof_$name -name "*.db" | head -1) > $O/${TAG}_${name}_rocprofv3_kernel_stats.txt 2>&1
  head -8 $O/${TAG}_${name}_rocprofv3_kernel_stats.txt | cut -c1-150
  find $O/prof_$name -name "*.db" -size +30M -delete
done
