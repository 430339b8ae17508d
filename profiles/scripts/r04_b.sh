# round 4, GPU session B: chunked open + device-side writer tests, the wave-scan test that failed on an over-strict assertion, then the
# default bench line and the same workload with the wave scan off
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; O=gpurun_out/r4b; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_partitioned.py tests/test_gpu_driver.py -m gpu -q --maxfail=10 -p no:cacheprovider -k "chunk or workspace_limit or index_write or index_decode or scanned_by_the_wave or partition or driver or golden or format1 or taxonomy" > $O/pytest_gpu.log 2>&1; tail -25 $O/pytest_gpu.log | cut -c1-300
timeout 1200 python bench.py --steps 10 --warmup 3 > $O/bench.json 2> $O/bench.log; echo "bench rc=$?"; grep -v "^$" $O/bench.log | tail -25 | cut -c1-600
timeout 300 env MTB_JOIN_COOP_MIN=100000000 python bench.py --steps 2 --warmup 1 --no-parity --no-legs > $O/bench_nocoop.json 2> $O/bench_nocoop.log; echo "nocoop rc=$?"; grep "stage ms" $O/bench_nocoop.log | cut -c1-300
