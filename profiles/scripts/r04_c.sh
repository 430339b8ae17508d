# round 4, GPU session C: the join's own-DNA bisection + LDS sort of deferred segments + index clone + prefetch (tests), the default
# bench line, then a database of 8 G targets written to and opened from files with the stand-alone driver on 60 M reads
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; O=gpurun_out/r4c; mkdir -p $O; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_partitioned.py tests/test_gpu_driver.py -m gpu -q --maxfail=10 -p no:cacheprovider -k "scanned_by_the_wave or long_runs or bench_ or clone or driver or many_matches or fused_batch or register_resident or two_bit" > $O/pytest_gpu.log 2>&1; tail -15 $O/pytest_gpu.log | cut -c1-300
timeout 600 python bench.py --steps 10 --warmup 3 > $O/bench.json 2> $O/bench.log; echo "bench rc=$?"; grep -v "^$" $O/bench.log | tail -14 | cut -c1-500
timeout 700 python profiles/scripts/e2e_big.py 8e9 60e6 64 2000000,4000000 > $O/e2e_big.txt 2>&1; echo "e2e rc=$?"; tail -16 $O/e2e_big.txt | cut -c1-600
