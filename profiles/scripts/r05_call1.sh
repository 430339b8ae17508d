# Round 5, GPU call 1: what the first hours of the round built, measured in one box session.
#   1. the GPU tests of the changed paths (k_score_many, the prefetch protocol, the reserve / epoch fix in the poison build)
#   2. bench.py, default workload, with the in-process A/B legs (k_score_many off; k_join_dir at q2w6 / q1w6 / q1w5) and the new held-out-organism leg
#   3. the gather probe (sorted vs unsorted directory lookups: VERDICT r4 item 7)
#   4. end to end on the 204 M-target database: driver defaults against --async-results 1
#   5. the partitioned path at N = 1 (after "self-exchange is a view")
# usage (GPU box): bash profiles/scripts/r05_call1.sh [TAG]; summaries stay under gpurun_out/<TAG>/
TAG=${1:-r05_c1}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; O=$R/gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/gather_probe profiles/scripts/gather_probe.hip > $O/build_probe.log 2>&1 ) &
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q --maxfail=5 -p no:cacheprovider -k "many_species or prefetched or many_matches or long_candidate_runs or fused_batch or zeroed_device_memory or runs_beyond_256" > $O/${TAG}_pytest_subset.log 2>&1; tail -n 3 $O/${TAG}_pytest_subset.log
wait
timeout 200 /tmp/gather_probe 16 1280 > $O/${TAG}_gather_probe.txt 2>&1; cat $O/${TAG}_gather_probe.txt
timeout 900 python bench.py --steps 5 --warmup 2 --ab "MTB_NO_SCORE_MANY=1;MTB_JOIN_VARIANT=q2w6;MTB_JOIN_VARIANT=q1w6;MTB_JOIN_VARIANT=q1w5" > $O/${TAG}_bench_default.json 2> $O/${TAG}_bench_default.log
grep "stage ms\|parity\|A/B\|leg \|library" $O/${TAG}_bench_default.log | cut -c1-260
E2E_VARIANTS="|--async-results 1" timeout 400 python profiles/scripts/e2e_big.py 2.04e8 60e6 64 2000000,4000000 > $O/${TAG}_e2e_204M_async_ab.txt 2>&1; grep "mtb_classify: 6\|max-reads" $O/${TAG}_e2e_204M_async_ab.txt | cut -c1-420
MTB_PART_TIMING=1 timeout 300 python bench.py --partitioned --reads 2000000 --targets 2e9 --species 24 --steps 5 --warmup 2 > $O/${TAG}_bench_partitioned_n1.json 2> $O/${TAG}_bench_partitioned_n1.log; tail -n 12 $O/${TAG}_bench_partitioned_n1.log | cut -c1-300
timeout 300 python bench.py --reads 2000000 --targets 2e9 --species 24 --steps 5 --warmup 2 --no-legs --no-cpu --cpu-reads 100000 > $O/${TAG}_bench_replicated_2M_2G.json 2> $O/${TAG}_bench_replicated_2M_2G.log; grep "stage ms" $O/${TAG}_bench_replicated_2M_2G.log | cut -c1-260
du -sh $O
