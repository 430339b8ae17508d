#!/bin/bash
# round 6, GPU call 17: window tiles -- three probes per search step (libmtb.so) against the bisection (libmtb_xr16.so = call 16's library), alternating processes;
# the tiers of the deferred reads on held-out genomes (MTB_MANY_VERBOSE); the whole GPU suite on the current state
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=$R/gpurun_out/r06_run17; mkdir -p $O; export TMPDIR=/tmp
for L in new r16 new r16; do
  if [ $L = r16 ]; then export MTB_LIB=$R/metabuli_amd/csrc/libmtb_xr16.so; else unset MTB_LIB; fi
  MTB_JOIN_VARIANT=window timeout 500 python bench.py --steps 10 --warmup 3 --no-legs --no-cpu --cpu-reads 100000 > $O/headline_$L.json 2>> $O/headline_$L.log
  echo "headline $L rc=$?"; grep -E "stage ms|parity" $O/headline_$L.log | tail -2 | cut -c1-220
done
unset MTB_LIB
MTB_MANY_VERBOSE=1 timeout 600 python bench.py --reads-from heldout --steps 2 --warmup 1 --no-legs --no-cpu --no-parity --ab "MTB_MANY_CAP=320" > $O/heldout.json 2> $O/heldout.log
echo "heldout rc=$?"; grep -E "stage ms|A/B|k_score_many|k_many_sort" $O/heldout.log | cut -c1-330 | tail -8
timeout 1500 python -m pytest tests -m gpu -x -q --timeout 400 > $O/pytest_gpu.txt 2>&1; tail -3 $O/pytest_gpu.txt
