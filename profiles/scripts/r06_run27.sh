#!/bin/bash
# round 6, GPU call 27: the sums of a run's first twelve candidates kept for the emission (libmtb.so) against the state of the final artifacts (libmtb_xfinal2.so), alternating processes
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=$R/gpurun_out/r06_run27; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "runs_of_a_dozen or long_candidate_runs or target_windows or match_and_sort or many_species or fused or a_few_long_reads" --timeout 300 > $O/pytest_subset.txt 2>&1; tail -2 $O/pytest_subset.txt
for L in new fin new fin; do
  if [ $L = fin ]; then export MTB_LIB=$R/metabuli_amd/csrc/libmtb_xfinal2.so; else unset MTB_LIB; fi
  timeout 600 python bench.py --reads-from heldout --steps 3 --warmup 1 --no-legs --no-cpu --cpu-reads 100000 > $O/heldout_$L.json 2>> $O/heldout_$L.log
  echo "heldout $L rc=$?"; grep -E "stage ms|parity" $O/heldout_$L.log | tail -2 | cut -c1-220
done
for L in new fin new fin; do
  if [ $L = fin ]; then export MTB_LIB=$R/metabuli_amd/csrc/libmtb_xfinal2.so; else unset MTB_LIB; fi
  MTB_JOIN_VARIANT=window timeout 500 python bench.py --steps 10 --warmup 3 --no-legs --no-cpu --cpu-reads 100000 > $O/headline_$L.json 2>> $O/headline_$L.log
  echo "headline $L rc=$?"; grep -E "stage ms|parity" $O/headline_$L.log | tail -2 | cut -c1-220
done
