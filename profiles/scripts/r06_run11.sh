#!/bin/bash
# round 6, GPU call 11: (1) the tighter candidate filter of the wave scan (h <= 2 x the lane's minimum so far) against round 5's (h <= 7), alternating
# processes; (2) full window tiles (256 queries) on the sparse short-read legs; (3) long reads with the window form by default.
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=$R/gpurun_out/r06_run11; mkdir -p $O; export TMPDIR=/tmp
AB="MTB_JOIN_VARIANT=window;MTB_JOIN_VARIANT=window,MTB_JOIN_WIN_QT=256;MTB_JOIN_VARIANT=q1w6"
for rep in 1 2; do
for lib in libmtb.so libmtb_xloose.so; do
  extra=""; [ "$rep$lib" = "1libmtb.so" ] && extra="--ab $AB"
  MTB_LIB=$R/metabuli_amd/csrc/$lib timeout 600 python bench.py --steps 5 --warmup 2 --no-parity $extra > $O/ab_${lib}_$rep.json 2> $O/ab_${lib}_$rep.log
  echo "== $lib run $rep rc=$?"; grep -E "stage ms|leg |A/B (paired|novel)" $O/ab_${lib}_$rep.log | cut -c1-170
done; done
timeout 600 python bench.py --seq-mode 3 --reads 200000 --read-len 10000 --steps 3 --warmup 1 --no-cpu --cpu-reads 333334 > $O/long_bench.json 2> $O/long_bench.log
echo "long rc=$?"; grep -E "stage ms|parity" $O/long_bench.log | cut -c1-200
timeout 600 python bench.py --reads-from heldout --steps 3 --warmup 1 --no-legs --no-parity > $O/heldout_tight.json 2> $O/heldout_tight.log; grep -E "stage ms" $O/heldout_tight.log | cut -c1-200
MTB_LIB=$R/metabuli_amd/csrc/libmtb_xloose.so timeout 600 python bench.py --reads-from heldout --steps 3 --warmup 1 --no-legs --no-parity > $O/heldout_loose.json 2> $O/heldout_loose.log; grep -E "stage ms" $O/heldout_loose.log | cut -c1-200
