# Round 5, GPU call 2: after the fixes of call 1 (stale bigidx for reads k_score_many took; one atomic per 16 claimed reads)
TAG=${1:-r05_c2}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; O=$R/gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q --maxfail=5 -p no:cacheprovider -k "many_species or prefetched or many_matches or long_candidate_runs or fused_batch or zeroed_device_memory or runs_beyond_256" > $O/${TAG}_pytest_subset.log 2>&1; tail -n 3 $O/${TAG}_pytest_subset.log | cut -c1-300
timeout 900 python bench.py --steps 5 --warmup 2 --ab "MTB_NO_SCORE_MANY=1;MTB_JOIN_VARIANT=q2w6;MTB_JOIN_VARIANT=q1w6;MTB_JOIN_VARIANT=q1w5" > $O/${TAG}_bench_default.json 2> $O/${TAG}_bench_default.log
grep "stage ms\|parity\|A/B\|leg \|library\|without" $O/${TAG}_bench_default.log | cut -c1-260
du -sh $O
