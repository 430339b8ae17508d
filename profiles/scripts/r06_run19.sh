#!/bin/bash
# round 6, GPU call 19: the wave-scan threshold (MTB_JOIN_COOP_MIN) again now that a scanned run costs a table + 19 instructions per 64 candidates: headline, pairs, long reads, one process each
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=$R/gpurun_out/r06_run19; mkdir -p $O; export TMPDIR=/tmp
AB="MTB_JOIN_COOP_MIN=8;MTB_JOIN_COOP_MIN=12;MTB_JOIN_COOP_MIN=16;MTB_JOIN_COOP_MIN=24"
timeout 600 python bench.py --steps 5 --warmup 3 --no-legs --no-cpu --no-parity --ab "$AB" > $O/headline.json 2> $O/headline.log
echo "headline rc=$?"; grep -E "A/B|stage ms" $O/headline.log | cut -c1-200
timeout 600 python bench.py --seq-mode 3 --reads 200000 --read-len 10000 --steps 2 --warmup 1 --no-cpu --no-parity --ab "$AB" > $O/long.json 2> $O/long.log
echo "long rc=$?"; grep -E "A/B|stage ms" $O/long.log | cut -c1-200
timeout 600 python bench.py --seq-mode 2 --reads 12500000 --steps 2 --warmup 1 --no-cpu --no-parity --ab "MTB_JOIN_COOP_MIN=8;MTB_JOIN_COOP_MIN=16" > $O/paired.json 2> $O/paired.log
echo "paired rc=$?"; grep -E "A/B|stage ms" $O/paired.log | cut -c1-200
