TAG=${1:-r05_c12}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; O=$R/gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q --maxfail=5 -p no:cacheprovider -k "target_windows or long_candidate_runs or runs_beyond_256 or fused_batch" > $O/${TAG}_pytest_subset.log 2>&1; tail -n 2 $O/${TAG}_pytest_subset.log | cut -c1-300
timeout 600 python bench.py --steps 5 --warmup 2 --no-legs --no-cpu --cpu-reads 200000 --ab "MTB_JOIN_WIN=0;MTB_JOIN_WIN_WG=4;MTB_JOIN_WIN_WG=10;MTB_JOIN_WIN_WG=20" > $O/${TAG}_bench_win.json 2> $O/${TAG}_bench_win.log; grep "stage ms\|parity\|A/B" $O/${TAG}_bench_win.log | cut -c1-200
