# Round 5: continuity with the earlier rounds' workloads: reads from 24 genomes on the uniform index (round 3's headline), 2400 genomes with uniform runs
TAG=${1:-r05_cont}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; O=$R/gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
timeout 300 python bench.py --species 24 --steps 10 --warmup 3 --no-legs --no-cpu --cpu-reads 100000 --ab "MTB_JOIN_WIN=0" > $O/${TAG}_bench_24genomes.json 2> $O/${TAG}_bench_24genomes.log; grep "stage ms\|parity\|A/B" $O/${TAG}_bench_24genomes.log | cut -c1-250
timeout 400 python bench.py --steps 5 --warmup 2 --no-conserved --no-legs --no-cpu --cpu-reads 100000 > $O/${TAG}_bench_uniform.json 2> $O/${TAG}_bench_uniform.log; grep "stage ms\|parity" $O/${TAG}_bench_uniform.log | cut -c1-250
