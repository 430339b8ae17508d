# Round 5: BASELINE.json configs[2] (200 k x 10 kb long reads) and configs[3]'s per-GPU shape (12.5 M read pairs) at FULL size on the heavy-tailed 16 G index:
# bench line with a parity sample, rocprofv3 kernel stats, FETCH_SIZE / WRITE_SIZE passes (profiles/pmc_traffic_diversity_mode<m>.json).
# usage (GPU box): bash profiles/scripts/r05_full_configs.sh [TAG] [modes, default "3 2"]
TAG=${1:-r05_full}
MODES=${2:-"3 2"}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; O=$R/gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
S=/tmp/mtb_prof_scratch
T=16000000000
for m in $MODES; do
  if [ $m = 3 ]; then ARGS="--seq-mode 3 --reads 200000 --read-len 10000"; NR=200000; RL=10000; NAME=long; CPUR=333334; else ARGS="--seq-mode 2 --reads 12500000"; NR=12500000; RL=150; NAME=paired; CPUR=16384; fi
  rm -rf $S; mkdir -p $S
  ( cd /tmp
    for pass in "d FETCH_SIZE" "e WRITE_SIZE"; do set -- $pass
      timeout 400 rocprofv3 --pmc $2 --kernel-trace --output-format csv -d $S/pmc_$1 -- python $R/bench.py $ARGS --steps 1 --warmup 1 --no-parity --no-legs > $O/pmc_${NAME}_$1.log 2>&1; echo "$NAME pmc pass $1 rc=$?"
    done )
  python profiles/scripts/pmc_summary.py $S/pmc_d $S/pmc_e > $O/${TAG}_${NAME}_pmc_counters.tsv 2> $O/pmc_summary_${NAME}.err
  python profiles/scripts/make_pmc_traffic.py $S/pmc $NR $RL $T $m "profiles/r05_final_${NAME}_pmc_counters.tsv (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of bench.py $ARGS --steps 1 --warmup 1 --no-parity --no-legs)" diversity_mode$m > $O/pmc_traffic_print_${NAME}.json 2> $O/pmc_traffic_${NAME}.err; cp profiles/pmc_traffic_diversity_mode$m.json $O/
  rm -rf $S/pmc_d $S/pmc_e
  ( cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d $S/prof_ks -o ks -- python $R/bench.py $ARGS --steps 3 --warmup 1 --no-parity --no-legs > $O/ks_${NAME}.json 2> $O/ks_${NAME}.log )
  python profiles/scripts/rocpd_summary.py $(find $S/prof_ks -name "*.db" | head -1) > $O/${TAG}_${NAME}_rocprofv3_kernel_stats.txt 2>&1; grep "k_join_dir\|k_seg_order\|k_score\|k_many\|k_ovf\|k_radix_scatter\|k_extract<2>" $O/${TAG}_${NAME}_rocprofv3_kernel_stats.txt | head -12 | cut -c1-150
  rm -rf $S
  timeout 700 python bench.py $ARGS --steps 3 --warmup 1 --no-cpu --cpu-reads $CPUR > $O/${TAG}_bench_${NAME}.json 2> $O/${TAG}_bench_${NAME}.log; echo "$NAME bench rc=$?"; grep "stage ms\|parity\|setup" $O/${TAG}_bench_${NAME}.log | cut -c1-250
done
du -sh $O
