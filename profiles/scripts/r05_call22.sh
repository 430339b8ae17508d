TAG=${1:-r05_c22}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; O=$R/gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
MTB_JOIN_VERBOSE=1 timeout 300 python bench.py --species 24 --steps 10 --warmup 5 --no-legs --no-cpu --cpu-reads 100000 > $O/${TAG}_bench_24genomes.json 2> $O/${TAG}_bench_24genomes.log; grep "stage ms\|parity\|tuned" $O/${TAG}_bench_24genomes.log | cut -c1-250
MTB_JOIN_VERBOSE=1 timeout 400 python bench.py --steps 10 --warmup 5 --no-legs --no-cpu --cpu-reads 100000 > $O/${TAG}_bench_default.json 2> $O/${TAG}_bench_default.log; grep "stage ms\|parity\|tuned" $O/${TAG}_bench_default.log | cut -c1-250
