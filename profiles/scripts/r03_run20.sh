cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r3t; O=gpurun_out/r3t
MTB_LIB=$PWD/metabuli_amd/csrc/libmtb_xlprof.so timeout 400 python bench.py --no-cpu --no-parity --steps 1 --warmup 1 --seq-mode 3 --reads 200000 --read-len 10000 > $O/bench_long_prof.json 2> $O/bench_long_prof.log; grep "k_score_long phases\|stage ms" $O/bench_long_prof.log | tail -5
