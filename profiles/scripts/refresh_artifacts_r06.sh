#!/bin/bash
# Regenerates round 6's final artifacts (what profiles/r06_final_* hold).  Only SUMMARIES are left under gpurun_out/<TAG>/.
# usage (GPU box): bash profiles/scripts/refresh_artifacts_r06.sh [TAG]
TAG=${1:-r06_final}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; O=$R/gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
S=/tmp/mtb_prof_scratch; rm -rf $S; mkdir -p $S
T=16000000000
# 1. HBM traffic of the step's kernels on the default workload: FETCH_SIZE and WRITE_SIZE in passes of their own (kernel trace only); the join's
#    instantiation pinned to what the timed run's tuner chooses on this workload (the window form)
( cd /tmp
  for pass in "d FETCH_SIZE" "e WRITE_SIZE"; do set -- $pass
    MTB_JOIN_VARIANT=window timeout 300 rocprofv3 --pmc $2 --kernel-trace --output-format csv -d $S/pmc_$1 -- python $R/bench.py --steps 1 --warmup 1 --no-parity --no-legs > $O/pmc_$1.log 2>&1; echo "pmc pass $1 rc=$?"
  done )
python profiles/scripts/pmc_summary.py $S/pmc_d $S/pmc_e > $O/${TAG}_pmc_counters.tsv 2> $O/pmc_summary.err
python profiles/scripts/make_pmc_traffic.py $S/pmc 10000000 150 $T 1 "profiles/${TAG}_pmc_counters.tsv (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of MTB_JOIN_VARIANT=window bench.py --steps 1 --warmup 1 --no-parity --no-legs)" diversity > $O/pmc_traffic_print.json 2> $O/pmc_traffic.err; cp profiles/pmc_traffic_diversity.json $O/pmc_traffic_diversity.json
rm -rf $S/pmc_d $S/pmc_e
# 2. rocprofv3 kernel stats of the same workload
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $S/prof_ks -o ks -- python $R/bench.py --steps 5 --warmup 2 --no-parity --no-legs > $O/ks_bench.json 2> $O/ks_bench.log )
python profiles/scripts/rocpd_summary.py $(find $S/prof_ks -name "*.db" | head -1) > $O/${TAG}_rocprofv3_kernel_stats.txt 2>&1; grep "k_join_dir\|k_join_tile\|k_score\|k_many\|k_ovf\|k_radix\|k_extract<2>\|k_big\|k_list" $O/${TAG}_rocprofv3_kernel_stats.txt | head -16 | cut -c1-150
rm -rf $S/prof_ks
# 3. the headline line (the driver's command; after the counter passes, so that it carries `traffic` / `effective` of this workload)
timeout 700 python bench.py --steps 20 --warmup 5 > $O/${TAG}_bench.json 2> $O/${TAG}_bench.log; echo "bench rc=$?"; grep "stage ms\|leg \|parity\|setup" $O/${TAG}_bench.log | cut -c1-250
cp bench_detail.json $O/${TAG}_bench_detail.json 2>/dev/null; wc -c $O/${TAG}_bench.json
# 4. configs[2] at full size: bench + kernel stats + FETCH / WRITE passes
timeout 600 python bench.py --seq-mode 3 --reads 200000 --read-len 10000 --steps 3 --warmup 1 --no-cpu --cpu-reads 333334 > $O/${TAG}_bench_long.json 2> $O/${TAG}_bench_long.log; echo "long bench rc=$?"; grep "stage ms\|parity" $O/${TAG}_bench_long.log | cut -c1-250
cp bench_detail.json $O/${TAG}_bench_long_detail.json 2>/dev/null
( cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d $S/prof_long -o ks -- python $R/bench.py --seq-mode 3 --reads 200000 --read-len 10000 --steps 3 --warmup 1 --no-parity > $O/ks_long.json 2> $O/ks_long.log )
python profiles/scripts/rocpd_summary.py $(find $S/prof_long -name "*.db" | head -1) > $O/${TAG}_long_rocprofv3_kernel_stats.txt 2>&1; head -12 $O/${TAG}_long_rocprofv3_kernel_stats.txt | cut -c1-150
rm -rf $S/prof_long
( cd /tmp
  for pass in "d FETCH_SIZE" "e WRITE_SIZE"; do set -- $pass
    timeout 400 rocprofv3 --pmc $2 --kernel-trace --output-format csv -d $S/pmcl_$1 -- python $R/bench.py --seq-mode 3 --reads 200000 --read-len 10000 --steps 1 --warmup 1 --no-parity > $O/pmcl_$1.log 2>&1; echo "long pmc pass $1 rc=$?"
  done )
python profiles/scripts/pmc_summary.py $S/pmcl_d $S/pmcl_e > $O/${TAG}_long_pmc_counters.tsv 2>> $O/pmc_summary.err
python profiles/scripts/make_pmc_traffic.py $S/pmcl 200000 10000 $T 3 "profiles/${TAG}_long_pmc_counters.tsv" diversity_mode3 > /dev/null 2>> $O/pmc_traffic.err; cp profiles/pmc_traffic_diversity_mode3.json $O/ 2>/dev/null
rm -rf $S/pmcl_d $S/pmcl_e
# 5. configs[3] per-GPU shape at full size
timeout 600 python bench.py --seq-mode 2 --reads 12500000 --steps 3 --warmup 1 --no-cpu --cpu-reads 200000 > $O/${TAG}_bench_paired.json 2> $O/${TAG}_bench_paired.log; echo "paired bench rc=$?"; grep "stage ms\|parity" $O/${TAG}_bench_paired.log | cut -c1-250
cp bench_detail.json $O/${TAG}_bench_paired_detail.json 2>/dev/null
# 6. reads of held-out genomes as the main workload: bench + kernel stats
( cd /tmp && timeout 500 rocprofv3 --kernel-trace --stats -d $S/prof_ho -o ks -- python $R/bench.py --reads-from heldout --steps 3 --warmup 1 --no-legs --no-cpu --cpu-reads 100000 > $O/${TAG}_bench_heldout.json 2> $O/${TAG}_bench_heldout.log )
grep "stage ms\|parity" $O/${TAG}_bench_heldout.log | cut -c1-250; cp /tmp/bench_detail.json $O/${TAG}_bench_heldout_detail.json 2>/dev/null      # (that run's working directory was /tmp)
python profiles/scripts/rocpd_summary.py $(find $S/prof_ho -name "*.db" | head -1) > $O/${TAG}_heldout_10M_rocprofv3_kernel_stats.txt 2>&1; head -14 $O/${TAG}_heldout_10M_rocprofv3_kernel_stats.txt | cut -c1-150
rm -rf $S
du -sh $O
# 7. the GPU suite on the final state
timeout 1500 python -m pytest tests -m gpu -q --timeout 400 > $O/${TAG}_pytest_gpu.txt 2>&1; tail -2 $O/${TAG}_pytest_gpu.txt
