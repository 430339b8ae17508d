#!/bin/bash
# round 6, GPU call 13: the wave scan's hamming sums from a per-query table in one register (ds_bpermute look-ups), uniform broadcasts by v_readlane;
# new library vs the previous one (libmtb_xprev.so) on reads of held-out genomes and on the headline
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=$R/gpurun_out/r06_run13; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "long_candidate_runs or target_windows or match_and_sort or many_species or deferred_reads_beyond or fused" --timeout 300 > $O/pytest_subset.txt 2>&1; tail -3 $O/pytest_subset.txt
if ! tail -1 $O/pytest_subset.txt | grep -q passed || tail -1 $O/pytest_subset.txt | grep -q failed; then
  echo "subset failed: rebuilding with exact ds_bpermute addresses"
  ( cd metabuli_amd/csrc && make libmtb_xexact.so X="-DMTB_BPERM_EXACT_ADDR" > $O/make_exact.log 2>&1 && cp libmtb_xexact.so libmtb.so )
  timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "long_candidate_runs or target_windows or match_and_sort" --timeout 300 > $O/pytest_subset_exact.txt 2>&1; tail -3 $O/pytest_subset_exact.txt
fi
for L in new prev new prev; do
  if [ $L = prev ]; then export MTB_LIB=$R/metabuli_amd/csrc/libmtb_xprev.so; else unset MTB_LIB; fi
  timeout 500 python bench.py --reads-from heldout --steps 3 --warmup 1 --no-legs --no-cpu --cpu-reads 100000 > $O/heldout_$L.json 2>> $O/heldout_$L.log
  echo "heldout $L rc=$?"; grep -E "stage ms|parity|tuner|join variant" $O/heldout_$L.log | tail -4 | cut -c1-220
done
unset MTB_LIB
timeout 500 python bench.py --reads-from heldout --steps 2 --warmup 1 --no-legs --no-cpu --no-parity --ab "MTB_JOIN_COOP_MIN=16;MTB_JOIN_COOP_MIN=8;MTB_JOIN_COOP_MIN=64" > $O/heldout_coop.json 2> $O/heldout_coop.log
grep -E "A/B |stage ms" $O/heldout_coop.log | cut -c1-200
for L in new prev; do
  if [ $L = prev ]; then export MTB_LIB=$R/metabuli_amd/csrc/libmtb_xprev.so; else unset MTB_LIB; fi
  timeout 500 python bench.py --steps 10 --warmup 3 --no-legs --no-cpu --cpu-reads 100000 > $O/headline_$L.json 2> $O/headline_$L.log
  echo "headline $L rc=$?"; grep -E "stage ms|parity|tuner" $O/headline_$L.log | tail -3 | cut -c1-220
done
