#!/bin/bash
# round 6, GPU call 29: continuity lines -- the earlier rounds' workloads on the final library (2400 genomes with uniform runs; 24 genomes with uniform runs = round 3's headline)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=$R/gpurun_out/r06_run29; mkdir -p $O; export TMPDIR=/tmp
timeout 600 python bench.py --no-conserved --steps 10 --warmup 3 --no-legs --no-cpu --cpu-reads 100000 > $O/r06_final_bench_uniform_runs.json 2> $O/uniform.log
echo "uniform rc=$?"; grep -E "stage ms|parity" $O/uniform.log | cut -c1-220
timeout 600 python bench.py --species 24 --steps 10 --warmup 3 --no-legs --no-cpu --cpu-reads 100000 > $O/r06_final_bench_24_genomes.json 2> $O/species24.log
echo "24 genomes rc=$?"; grep -E "stage ms|parity" $O/species24.log | cut -c1-220
