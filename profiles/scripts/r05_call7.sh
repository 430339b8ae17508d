# Round 5, GPU call 7: the window-staged join (k_join_dir<.., WIN>) against the sector-random one, in-process
TAG=${1:-r05_c7}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; O=$R/gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q --maxfail=5 -p no:cacheprovider -k "target_windows or many_species or many_matches or long_candidate_runs or fused_batch or runs_beyond_256" > $O/${TAG}_pytest_subset.log 2>&1; tail -n 3 $O/${TAG}_pytest_subset.log | cut -c1-300
timeout 900 python bench.py --steps 5 --warmup 2 --ab "MTB_JOIN_WIN=0;MTB_JOIN_WIN_QT=52;MTB_JOIN_WIN_QT=56;MTB_JOIN_WIN_QT=60;MTB_JOIN_WIN_QT=64" --no-cpu > $O/${TAG}_bench_default.json 2> $O/${TAG}_bench_default.log
grep "stage ms\|parity\|A/B\|leg \|without" $O/${TAG}_bench_default.log | cut -c1-260
