#!/bin/bash
# round 6, GPU call 7: hamming sums from the codon-pair table (4 byte look-ups) against the nibble rows (A/B build libmtb_xnoh2.so), alternating processes on one
# box: headline + legs; then the whole GPU suite.
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=$R/gpurun_out/r06_run7; mkdir -p $O; export TMPDIR=/tmp
for rep in 1 2; do
for lib in libmtb.so libmtb_xnoh2.so; do
  MTB_LIB=$R/metabuli_amd/csrc/$lib timeout 600 python bench.py --steps 5 --warmup 2 --no-parity > $O/ab_${lib}_$rep.json 2> $O/ab_${lib}_$rep.log
  echo "== $lib run $rep rc=$?"; grep -E "stage ms|leg " $O/ab_${lib}_$rep.log | cut -c1-160
done; done
timeout 1500 python -m pytest tests -m gpu -q -x > $O/pytest_gpu.txt 2>&1; tail -5 $O/pytest_gpu.txt
du -sh $O
