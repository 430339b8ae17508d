# round 4, GPU session G: striped overflow list -- tests, the default bench line, and the same with the join compiled for 5 waves per SIMD
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; O=gpurun_out/r4g; mkdir -p $O; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_partitioned.py -m gpu -q --maxfail=10 -p no:cacheprovider -k "long_reads_do_not or many_matches or scanned_by_the_wave or fused_batch or register_resident or long_runs or epoch or hbm_budgeted or two_streams or bench_" > $O/pytest_gpu.log 2>&1; tail -6 $O/pytest_gpu.log | cut -c1-300
timeout 600 python bench.py --steps 10 --warmup 3 > $O/bench.json 2> $O/bench.log; echo "bench rc=$?"; grep "stage ms\|leg \|parity" $O/bench.log | cut -c1-300
MTB_LIB=$R/metabuli_amd/csrc/libmtb_xw5.so timeout 400 python bench.py --steps 5 --warmup 2 --no-parity --no-legs > $O/bench_w5.json 2> $O/bench_w5.log; echo "w5 rc=$?"; grep "stage ms" $O/bench_w5.log | cut -c1-300
