# round 4, GPU session D: the whole GPU suite on the one-bisection join, the default bench line, the same index without conserved segments
# (uniform candidate runs: the round-3 diversity workload), the 8 G-target database from files with the driver
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; O=gpurun_out/r4d; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q --maxfail=10 -p no:cacheprovider --deselect tests/test_gpu_parity.py::test_database_opens_chunk_by_chunk > $O/pytest_gpu.log 2>&1; tail -12 $O/pytest_gpu.log | cut -c1-300
timeout 600 python bench.py --steps 10 --warmup 3 > $O/bench.json 2> $O/bench.log; echo "bench rc=$?"; grep -v "^$" $O/bench.log | tail -12 | cut -c1-400
timeout 400 python bench.py --steps 5 --warmup 2 --no-conserved --no-legs --no-cpu --cpu-reads 100000 > $O/bench_uniform.json 2> $O/bench_uniform.log; echo "uniform rc=$?"; grep "stage ms\|parity" $O/bench_uniform.log | cut -c1-300
timeout 600 python profiles/scripts/e2e_big.py 8e9 60e6 64 4000000 > $O/e2e_big.txt 2>&1; echo "e2e rc=$?"; grep -v "^[0-9. ]*\t" $O/e2e_big.txt | head -12 | cut -c1-600
