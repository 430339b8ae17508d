# regenerates the round's headline artifacts under gpurun_out/ (copy the ones to keep into profiles/)
# usage: bash profiles/scripts/refresh_artifacts.sh TAG
set -x
TAG=${1:-final}
cd /root/repo
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
python bench.py --steps 3 --warmup 1 > $O/bench_$TAG.json 2> $O/bench_$TAG.log
tail -2 $O/bench_$TAG.log
rm -rf $O/prof_ks && mkdir -p $O/prof_ks
rocprofv3 --kernel-trace --stats -d $O/prof_ks -o ks -- python bench.py --steps 2 --warmup 1 --no-cpu > $O/prof_ks/bench.json 2> $O/prof_ks/bench.log
python profiles/scripts/rocpd_summary.py $(find $O/prof_ks -name "*.db" | head -1) > $O/${TAG}_kernel_stats.txt
head -12 $O/${TAG}_kernel_stats.txt
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf $O/prof_pmc_$C && mkdir -p $O/prof_pmc_$C
  rocprofv3 --pmc $C --output-format csv -d $O/prof_pmc_$C -o pmc -- python bench.py --steps 1 --warmup 0 --no-cpu > /dev/null 2> $O/prof_pmc_$C/bench.log
done
( echo "# rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), bench.py --steps 1 --warmup 0 --no-cpu; KB per launch (avg, launches)";
  for C in FETCH_SIZE WRITE_SIZE; do python profiles/scripts/pmc_summary.py $(dirname $(find $O/prof_pmc_$C -name "*counter_collection.csv" | head -1)); done ) > $O/${TAG}_pmc_hbm.txt
cat $O/${TAG}_pmc_hbm.txt | cut -c1-200
python bench.py --no-cpu --steps 2 --seq-mode 2 --reads 5000000 > $O/bench_${TAG}_paired.json 2> $O/bench_${TAG}_paired.log; grep "stage ms" $O/bench_${TAG}_paired.log
python bench.py --no-cpu --steps 2 --seq-mode 3 --reads 50000 --read-len 10000 --targets 2e9 > $O/bench_${TAG}_long.json 2> $O/bench_${TAG}_long.log; grep "stage ms" $O/bench_${TAG}_long.log
