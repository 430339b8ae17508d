# round 4, GPU session A: the GPU suite (new: wave-scanned candidate runs, partitioned pairs / long reads / fallback join, bench in small),
# then the default bench line (2400 genomes, heavy-tailed runs, legs of the other configurations), then the same workload with the
# wave scan switched off (A/B for the notes; bounded by a timeout: a lane walks runs of thousands on its own there)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; O=gpurun_out/r4a; mkdir -p $O; export TMPDIR=/tmp
timeout 1700 python -m pytest tests -m gpu -q --maxfail=10 -p no:cacheprovider > $O/pytest_gpu.log 2>&1; tail -25 $O/pytest_gpu.log | cut -c1-300
timeout 1200 python bench.py --steps 10 --warmup 3 > $O/bench.json 2> $O/bench.log; echo "bench rc=$?"; grep -v "^$" $O/bench.log | tail -25 | cut -c1-400
timeout 400 env MTB_JOIN_COOP_MIN=100000000 python bench.py --steps 2 --warmup 1 --no-parity --no-legs > $O/bench_nocoop.json 2> $O/bench_nocoop.log; echo "nocoop rc=$?"; grep "stage ms" $O/bench_nocoop.log | cut -c1-300
