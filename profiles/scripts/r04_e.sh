# round 4, GPU session E: the two corrected tests, then where k_join_dir spends its cycles on the heavy-tailed workload (profiling build)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; O=gpurun_out/r4e; mkdir -p $O; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q --maxfail=10 -p no:cacheprovider -k "long_reads_do_not or many_matches or scanned_by_the_wave or fused_batch or register_resident" > $O/pytest_gpu.log 2>&1; tail -6 $O/pytest_gpu.log | cut -c1-300
MTB_LIB=$R/metabuli_amd/csrc/libmtb_prof.so timeout 400 python bench.py --steps 3 --warmup 1 --no-parity --no-legs > $O/bench_prof.json 2> $O/bench_prof.log; echo "prof rc=$?"; grep "stage ms\|phase cycles" $O/bench_prof.log | cut -c1-500
MTB_LIB=$R/metabuli_amd/csrc/libmtb_prof.so timeout 400 python bench.py --steps 3 --warmup 1 --no-parity --no-legs --no-conserved > $O/bench_prof_uniform.json 2> $O/bench_prof_uniform.log; echo "prof uniform rc=$?"; grep "stage ms\|phase cycles" $O/bench_prof_uniform.log | cut -c1-500
