# First GPU call of the next round: what the last hours of round 4 built WITHOUT a GPU (the budget was spent), measured.
#   1. the GPU suite (the emulated run -- tests/hipemu -- passed; this is the real thing)
#   2. end to end on the 204 M-target database from files: driver defaults against --async-results 1 (results copied out while the next batch
#      computes), every run into an empty output directory, rows checksummed
#   3. the 24-genome workload with the join at 5 and at 6 waves per SIMD (round 4's regression on that workload: 53.9 ms against 42 - 45)
# usage (GPU box): bash profiles/scripts/next_round_first.sh [TAG]; only summaries stay under gpurun_out/<TAG>/
TAG=${1:-r05_first}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; O=$R/gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
( cd metabuli_amd/csrc && make libmtb_xw6.so X="-DMTB_JOIN_WAVES=6" > $O/build_w6.log 2>&1; make libmtb_xq1w6.so X="-DMTB_JOIN_DIR_QPT=1 -DMTB_JOIN_WAVES=6" > $O/build_q1w6.log 2>&1 ) &
# (one query per thread at 6 waves per SIMD: 80 registers without a spill -- two queries per thread at 6 waves spill 16; logic checked on the emulated build)
timeout 1500 python -m pytest tests -m gpu -q --maxfail=10 -p no:cacheprovider > $O/${TAG}_pytest_gpu.log 2>&1; tail -n 3 $O/${TAG}_pytest_gpu.log
E2E_VARIANTS="|--async-results 1" timeout 300 python profiles/scripts/e2e_big.py 2.04e8 60e6 64 2000000,4000000 > $O/${TAG}_e2e_204M_async_ab.txt 2>&1; grep "mtb_classify: 6\|max-reads" $O/${TAG}_e2e_204M_async_ab.txt | cut -c1-360
wait
timeout 300 python bench.py --species 24 --steps 10 --warmup 3 --no-legs --no-cpu --cpu-reads 100000 > $O/${TAG}_bench_24genomes_w5.json 2> $O/${TAG}_bench_24genomes_w5.log; grep "stage ms\|parity" $O/${TAG}_bench_24genomes_w5.log | cut -c1-250
MTB_LIB=$R/metabuli_amd/csrc/libmtb_xw6.so timeout 300 python bench.py --species 24 --steps 10 --warmup 3 --no-legs --no-cpu --cpu-reads 100000 > $O/${TAG}_bench_24genomes_w6.json 2> $O/${TAG}_bench_24genomes_w6.log; grep "stage ms\|parity" $O/${TAG}_bench_24genomes_w6.log | cut -c1-250
MTB_LIB=$R/metabuli_amd/csrc/libmtb_xq1w6.so timeout 300 python bench.py --species 24 --steps 10 --warmup 3 --no-legs --no-cpu --cpu-reads 100000 > $O/${TAG}_bench_24genomes_q1w6.json 2> $O/${TAG}_bench_24genomes_q1w6.log; grep "stage ms\|parity" $O/${TAG}_bench_24genomes_q1w6.log | cut -c1-250
MTB_LIB=$R/metabuli_amd/csrc/libmtb_xq1w6.so timeout 400 python bench.py --steps 5 --warmup 2 --no-legs --no-cpu --cpu-reads 100000 > $O/${TAG}_bench_default_q1w6.json 2> $O/${TAG}_bench_default_q1w6.log; grep "stage ms\|parity" $O/${TAG}_bench_default_q1w6.log | cut -c1-250
du -sh $O
