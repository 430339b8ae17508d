#!/bin/bash
# round 6, GPU call 21: the tail places of the candidates read ahead reserved by one early atomic (libmtb.so) against the state of the final artifacts (libmtb_xfinal1.so), alternating processes
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=$R/gpurun_out/r06_run21; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "long_candidate_runs or target_windows or match_and_sort or many_species or deferred_reads_beyond or fused or a_few_long_reads" --timeout 300 > $O/pytest_subset.txt 2>&1; tail -2 $O/pytest_subset.txt
for L in new fin new fin; do
  if [ $L = fin ]; then export MTB_LIB=$R/metabuli_amd/csrc/libmtb_xfinal1.so; else unset MTB_LIB; fi
  MTB_JOIN_VARIANT=window timeout 500 python bench.py --steps 10 --warmup 3 --no-legs --no-cpu --cpu-reads 100000 > $O/headline_$L.json 2>> $O/headline_$L.log
  echo "headline $L rc=$?"; grep -E "stage ms|parity" $O/headline_$L.log | tail -2 | cut -c1-220
done
unset MTB_LIB
timeout 600 python bench.py --seq-mode 2 --reads 12500000 --steps 2 --warmup 1 --no-cpu --cpu-reads 100000 > $O/paired.json 2> $O/paired.log
echo "paired rc=$?"; grep -E "stage ms|parity" $O/paired.log | cut -c1-200
