#!/bin/bash
# round 6, GPU call 23: longer tails of the slot segments (MTB_TAIL_MIN: fewer reads overflow into the many-species path, more slots to clear / read) -- pairs, held-out reads, headline, one process each
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=$R/gpurun_out/r06_run23; mkdir -p $O; export TMPDIR=/tmp
timeout 700 python bench.py --seq-mode 2 --reads 12500000 --steps 2 --warmup 1 --no-cpu --no-parity --ab "MTB_TAIL_MIN=64;MTB_TAIL_MIN=96" > $O/paired.json 2> $O/paired.log
echo "paired rc=$?"; grep -E "A/B|stage ms" $O/paired.log | cut -c1-200
timeout 700 python bench.py --reads-from heldout --steps 2 --warmup 1 --no-legs --no-cpu --no-parity --ab "MTB_TAIL_MIN=32;MTB_TAIL_MIN=64" > $O/heldout.json 2> $O/heldout.log
echo "heldout rc=$?"; grep -E "A/B|stage ms" $O/heldout.log | cut -c1-200
timeout 700 python bench.py --steps 5 --warmup 3 --no-legs --no-cpu --no-parity --ab "MTB_TAIL_MIN=32;MTB_TAIL_MIN=48" > $O/headline.json 2> $O/headline.log
echo "headline rc=$?"; grep -E "A/B|stage ms" $O/headline.log | cut -c1-200
