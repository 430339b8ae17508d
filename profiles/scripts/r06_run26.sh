#!/bin/bash
# round 6, GPU call 26: HBM traffic of the kernels on 10 M reads of held-out genomes (FETCH_SIZE / WRITE_SIZE passes of their own), then that workload's bench line with `traffic` / `effective`
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=$R/gpurun_out/r06_run26; mkdir -p $O; S=/tmp/mtb_pmc_ho; rm -rf $S; mkdir -p $S; export TMPDIR=/tmp
( cd /tmp
  for pass in "d FETCH_SIZE" "e WRITE_SIZE"; do set -- $pass
    timeout 400 rocprofv3 --pmc $2 --kernel-trace --output-format csv -d $S/pmc_$1 -- python $R/bench.py --reads-from heldout --steps 1 --warmup 1 --no-parity --no-legs --no-cpu > $O/pmc_$1.log 2>&1; echo "heldout pmc pass $1 rc=$?"
  done )
python profiles/scripts/pmc_summary.py $S/pmc_d $S/pmc_e > $O/r06_final_heldout_pmc_counters.tsv 2> $O/pmc_summary.err
python profiles/scripts/make_pmc_traffic.py $S/pmc 10000000 150 16000000000 1 "profiles/r06_final_heldout_pmc_counters.tsv (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of bench.py --reads-from heldout --steps 1 --warmup 1 --no-parity --no-legs --no-cpu)" diversity_heldout > $O/pmc_traffic_print.json 2> $O/pmc_traffic.err
cp profiles/pmc_traffic_diversity_heldout.json $O/ 2>/dev/null; rm -rf $S
timeout 600 python bench.py --reads-from heldout --steps 3 --warmup 1 --no-legs --no-cpu --cpu-reads 100000 > $O/r06_final_bench_heldout.json 2> $O/r06_final_bench_heldout.log
echo "heldout bench rc=$?"; grep -E "stage ms|parity" $O/r06_final_bench_heldout.log | cut -c1-220; cp bench_detail.json $O/r06_final_bench_heldout_detail.json
grep "k_join_dir" $O/r06_final_heldout_pmc_counters.tsv | cut -c1-200
