#!/bin/bash
# round 6, GPU call 18: why reads leave k_score_fast on the headline batch (debugging build), where the join's cycles go now (profiling build: headline and held-out reads)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=$R/gpurun_out/r06_run18; mkdir -p $O; export TMPDIR=/tmp
MTB_LIB=$R/metabuli_amd/csrc/libmtb_dbg.so timeout 500 python bench.py --steps 2 --warmup 1 --no-legs --no-cpu --no-parity > $O/dbg.json 2> $O/dbg.log
echo "dbg rc=$?"; grep -E "k_score_fast exits|stage ms" $O/dbg.log | cut -c1-700
MTB_JOIN_VARIANT=window MTB_LIB=$R/metabuli_amd/csrc/libmtb_prof.so timeout 500 python bench.py --steps 2 --warmup 1 --no-legs --no-cpu --no-parity > $O/prof.json 2> $O/prof.log
echo "prof rc=$?"; grep -E "phase cycles|stage ms" $O/prof.log | cut -c1-700
MTB_LIB=$R/metabuli_amd/csrc/libmtb_prof.so timeout 500 python bench.py --reads-from heldout --steps 2 --warmup 1 --no-legs --no-cpu --no-parity > $O/prof_heldout.json 2> $O/prof_heldout.log
echo "prof heldout rc=$?"; grep -E "phase cycles|stage ms" $O/prof_heldout.log | cut -c1-700
