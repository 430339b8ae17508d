# First GPU call of the next round (about 3 minutes of box time): what round 3 could not run any more.
#   1. the long-read configuration with its 20 000-read parity sample on the last change of round 3 (k_score_long's combination phase,
#      64 candidate paths per step) -> gpurun_out/next/bench_long.json must say parity mismatches 0 twice;
#   2. rocprofv3 kernel stats of the same configuration (k_score_long was 53.4 ms before that change, 42.4 ms in the one run after it);
#   3. the GPU suite.
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; O=gpurun_out/next; mkdir -p $O; export TMPDIR=/tmp
timeout 600 python bench.py --cpu-reads 20000 --cpu-targets 16e6 --steps 2 --warmup 2 --seq-mode 3 --reads 200000 --read-len 10000 > $O/bench_long.json 2> $O/bench_long.log
grep "stage ms\|parity" $O/bench_long.log | cut -c1-200
rm -rf $O/prof_long && mkdir -p $O/prof_long
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $R/$O/prof_long -o ks -- python $R/bench.py --steps 3 --warmup 2 --no-parity --seq-mode 3 --reads 200000 --read-len 10000 > $R/$O/prof_long/bench.json 2> $R/$O/prof_long/bench.log )
python profiles/scripts/rocpd_summary.py $(find $O/prof_long -name "*.db" | head -1) > $O/long_rocprofv3_kernel_stats.txt 2>&1; head -8 $O/long_rocprofv3_kernel_stats.txt | cut -c1-150
find $O/prof_long -name "*.db" -size +30M -delete
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; tail -3 $O/pytest_gpu.log
