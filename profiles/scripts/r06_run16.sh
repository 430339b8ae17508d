#!/bin/bash
# round 6, GPU call 16: window tiles -- run ends from eight words read at once, evaluation from four words read at once (sums kept for the emission); new library vs the
# previous one (libmtb_xprev.so = the state of call 15), alternating processes on one box
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=$R/gpurun_out/r06_run16; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "long_candidate_runs or target_windows or match_and_sort or many_species or deferred_reads_beyond or fused" --timeout 300 > $O/pytest_subset.txt 2>&1; tail -3 $O/pytest_subset.txt
for L in new prev new prev; do
  if [ $L = prev ]; then export MTB_LIB=$R/metabuli_amd/csrc/libmtb_xprev.so; else unset MTB_LIB; fi
  MTB_JOIN_VARIANT=window timeout 500 python bench.py --steps 10 --warmup 3 --no-legs --no-cpu --cpu-reads 100000 > $O/headline_$L.json 2>> $O/headline_$L.log
  echo "headline $L rc=$?"; grep -E "stage ms|parity" $O/headline_$L.log | tail -2 | cut -c1-220
done
unset MTB_LIB
timeout 600 python bench.py --reads-from heldout --steps 3 --warmup 1 --no-legs --no-cpu --cpu-reads 100000 > $O/heldout.json 2> $O/heldout.log
echo "heldout rc=$?"; grep -E "stage ms|parity" $O/heldout.log | cut -c1-260 | tail -3
