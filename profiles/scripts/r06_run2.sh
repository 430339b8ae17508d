#!/bin/bash
# round 6, GPU call 2: (1) in-process A/B of the join's forms on the headline batch: window with / without pre-bounded tile windows, sector-random,
# the wave-scan threshold (divergence of the per-lane candidate loops); (2) SQ counter passes of the window join (is it issue-bound?).
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=gpurun_out/r06_run2; mkdir -p $O
AB="MTB_JOIN_VARIANT=window;MTB_JOIN_VARIANT=window,MTB_JOIN_NO_PREWIN=1;MTB_JOIN_VARIANT=q1w6;MTB_JOIN_VARIANT=window,MTB_JOIN_COOP_MIN=8;MTB_JOIN_VARIANT=window,MTB_JOIN_COOP_MIN=16;MTB_JOIN_VARIANT=window,MTB_JOIN_COOP_MIN=64;MTB_JOIN_VARIANT=q1w6,MTB_JOIN_COOP_MIN=8;MTB_JOIN_VARIANT=q1w6,MTB_JOIN_COOP_MIN=16;MTB_JOIN_VARIANT=window,MTB_JOIN_WIN_QT=224;MTB_JOIN_VARIANT=window;MTB_JOIN_VARIANT=window,MTB_JOIN_NO_PREWIN=1"
timeout 1200 python bench.py --steps 5 --warmup 2 --no-cpu --cpu-reads 200000 --ab "$AB" > $O/bench_ab.json 2> $O/bench_ab.log
grep -E "A/B|stage ms|leg " $O/bench_ab.log
cp bench_detail.json $O/bench_ab_detail.json 2>/dev/null
cd /tmp && export TMPDIR=/tmp
run() { d=$1; shift; c="$1"; shift
  MTB_JOIN_VARIANT=window timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/$O/pmc_$d -- python $R/bench.py --steps 1 --warmup 1 --no-parity --no-legs > $R/$O/pmc_$d.log 2>&1
  echo "pass $d rc=$?"; }
run a "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VALU"
run b "SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA"
run c "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_FLAT SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE GRBM_COUNT"
cd $R
python profiles/scripts/pmc_summary.py $O/pmc_a $O/pmc_b $O/pmc_c > $O/pmc_sq_summary.tsv 2>&1
find $O -name "*counter_collection.csv" -size +20M -delete; find $O -name "*.db" -size +20M -delete
python - <<PY
import csv
rows=list(csv.reader(open("$O/pmc_sq_summary.tsv"),delimiter="\t"))
for r in rows[1:]:
    if any(k in r[0] for k in ("join_dir","extract<2","score_fast","radix_scatter","score_many","k_score<")):
        print(r[0]); print("   "+"  ".join(f"{h}={v}" for h,v in zip(rows[0][1:],r[1:])))
PY
