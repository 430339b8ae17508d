#!/bin/bash
# round 6, GPU call 14: why reads of held-out genomes / pairs / long reads take two sub-batches (budget diagnostics), where the held-out join's cycles go
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=$R/gpurun_out/r06_run14; mkdir -p $O; export TMPDIR=/tmp
export MTB_HOST_TIMING=1
MTB_LIB=$R/metabuli_amd/csrc/libmtb_prof.so timeout 500 python bench.py --reads-from heldout --steps 2 --warmup 1 --no-legs --no-cpu --no-parity > $O/heldout_prof.json 2> $O/heldout_prof.log
echo "heldout prof rc=$?"; grep -E "mtb budget|stage ms|phase cycles" $O/heldout_prof.log | cut -c1-600 | head -12
timeout 500 python bench.py --seq-mode 2 --reads 12500000 --steps 2 --warmup 1 --no-cpu --no-parity > $O/paired.json 2> $O/paired.log
echo "paired rc=$?"; grep -E "mtb budget|stage ms" $O/paired.log | cut -c1-600 | head -10
timeout 500 python bench.py --seq-mode 3 --reads 200000 --read-len 10000 --steps 2 --warmup 1 --no-cpu --no-parity > $O/long.json 2> $O/long.log
echo "long rc=$?"; grep -E "mtb budget|stage ms" $O/long.log | cut -c1-600 | head -10
