#!/bin/bash
# round 6, GPU call 20: the LDS bitonic network with wavefront fences between its wave-local steps (k_many_sort, k_seg_order), the exact-segment tier fed from the grouped
# overflow list, wave-scan threshold 12 -- new library vs call 16's (libmtb_xr16.so), alternating processes
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=$R/gpurun_out/r06_run20; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "many_species or deferred_reads_beyond or sync_long or sync_xlong or long_candidate_runs or a_few_long_reads" --timeout 300 > $O/pytest_subset.txt 2>&1; tail -3 $O/pytest_subset.txt
for L in new r16 new r16; do
  if [ $L = r16 ]; then export MTB_LIB=$R/metabuli_amd/csrc/libmtb_xr16.so; else unset MTB_LIB; fi
  timeout 600 python bench.py --reads-from heldout --steps 3 --warmup 1 --no-legs --no-cpu --cpu-reads 100000 > $O/heldout_$L.json 2>> $O/heldout_$L.log
  echo "heldout $L rc=$?"; grep -E "stage ms|parity" $O/heldout_$L.log | tail -2 | cut -c1-220
  cp bench_detail.json $O/heldout_${L}_detail.json
done
for L in new r16; do
  if [ $L = r16 ]; then export MTB_LIB=$R/metabuli_amd/csrc/libmtb_xr16.so; else unset MTB_LIB; fi
  timeout 600 python bench.py --seq-mode 3 --reads 200000 --read-len 10000 --steps 3 --warmup 1 --no-cpu --cpu-reads 333334 > $O/long_$L.json 2>> $O/long_$L.log
  echo "long $L rc=$?"; grep -E "stage ms|parity" $O/long_$L.log | tail -2 | cut -c1-220
done
