set -x
cd /root/repo
export TMPDIR=/tmp
python bench.py --steps 3 --warmup 1 > gpurun_out/bench_r01_final.json 2> gpurun_out/bench_r01_final.log
tail -3 gpurun_out/bench_r01_final.log
rm -rf gpurun_out/prof_ks && mkdir -p gpurun_out/prof_ks
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_ks -o ks -- python bench.py --steps 2 --warmup 1 --no-cpu > gpurun_out/prof_ks/bench.json 2> gpurun_out/prof_ks/bench.log
find gpurun_out/prof_ks -name "*.db" | head
DB=$(find gpurun_out/prof_ks -name "*.db" | head -1)
python profiles/scripts/rocpd_summary.py $DB > gpurun_out/r01_final2_kernel_stats.txt
head -30 gpurun_out/r01_final2_kernel_stats.txt
