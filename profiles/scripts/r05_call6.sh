# Round 5, GPU call 6: k_score_long with the lane walk of tiny blocks: GPU tests of the long-read and many-species paths, default bench with legs, kernel trace of the held-out workload
TAG=${1:-r05_c6}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; O=$R/gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q --maxfail=5 -p no:cacheprovider -k "many_species or many_matches or long_candidate_runs or fused_batch or golden or runs_beyond_256 or long" > $O/${TAG}_pytest_subset.log 2>&1; tail -n 3 $O/${TAG}_pytest_subset.log | cut -c1-300
timeout 900 python bench.py --steps 5 --warmup 2 > $O/${TAG}_bench_default.json 2> $O/${TAG}_bench_default.log
grep "stage ms\|parity\|A/B\|leg \|without" $O/${TAG}_bench_default.log | cut -c1-260
S=/tmp/mtb_prof_scratch; rm -rf $S; mkdir -p $S
( cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d $S/prof_ks -o ks -- python $R/bench.py --reads-from heldout --reads 2000000 --steps 3 --warmup 1 --no-legs --no-parity > $O/${TAG}_novel_ks.json 2> $O/${TAG}_novel_ks.log )
python profiles/scripts/rocpd_summary.py $(find $S/prof_ks -name "*.db" | head -1) > $O/${TAG}_novel_rocprofv3_kernel_stats.txt 2>&1; grep "k_score\|k_many\|k_ovf\|k_join_dir\|k_big\|k_segsort" $O/${TAG}_novel_rocprofv3_kernel_stats.txt | head -14 | cut -c1-150
rm -rf $S
