# Round 5, GPU call 4: k_many_sort + k_score_long<4096, 1024> for the reads beyond k_score_many's staging; two sort passes instead of three as an A/B leg
TAG=${1:-r05_c4}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; O=$R/gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q --maxfail=5 -p no:cacheprovider -k "many_species or prefetched or many_matches or long_candidate_runs or fused_batch or zeroed_device_memory or runs_beyond_256" > $O/${TAG}_pytest_subset.log 2>&1; tail -n 3 $O/${TAG}_pytest_subset.log | cut -c1-300
MTB_MANY_VERBOSE=1 timeout 900 python bench.py --steps 5 --warmup 2 --ab "MTB_NO_MANY_SORT=1;MTB_SORT_PAIRS=2" > $O/${TAG}_bench_default.json 2> $O/${TAG}_bench_default.log
grep "stage ms\|parity\|A/B\|leg \|library\|without" $O/${TAG}_bench_default.log | cut -c1-260; grep "k_many_sort" $O/${TAG}_bench_default.log | sort | uniq -c | cut -c1-200
du -sh $O
