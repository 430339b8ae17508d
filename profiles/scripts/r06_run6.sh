#!/bin/bash
# round 6, GPU call 6: the consolidated join (one window form, the tuner's third candidate) + the inter-process hand-over of the index on real hardware;
# SQ counters of the join on reads of held-out genomes.
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=$R/gpurun_out/r06_run6; mkdir -p $O; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "windows_staged or share_a_long or resident_index_handed or two_ranks_share or test_bench_path or heavy_tailed_workload" > $O/pytest_subset.txt 2>&1; tail -4 $O/pytest_subset.txt
timeout 900 python bench.py --steps 5 --warmup 2 --no-cpu --cpu-reads 200000 > $O/bench.json 2> $O/bench.log
echo "bench rc=$?"; grep -E "stage ms|leg |parity|headline line" $O/bench.log | cut -c1-200; cat $O/bench.json | cut -c1-1500
cp bench_detail.json $O/bench_detail.json 2>/dev/null
S=/tmp/mtb_prof_scratch; rm -rf $S; mkdir -p $S
cd /tmp
run() { d=$1; shift; c="$1"; shift
  timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $S/pmc_$d -- python $R/bench.py --reads-from heldout --reads 4000000 --steps 1 --warmup 1 --no-parity --no-legs > $O/pmc_$d.log 2>&1
  echo "pass $d rc=$?"; }
run a "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VALU"
run b "SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_THREAD_CYCLES_VALU"
cd $R
python profiles/scripts/pmc_summary.py $S/pmc_a $S/pmc_b > $O/r06_heldout_4M_pmc_sq.tsv 2>&1
rm -rf $S
python - <<PY
import csv
rows=list(csv.reader(open("$O/r06_heldout_4M_pmc_sq.tsv"),delimiter="\t"))
for r in rows[1:]:
    if any(k in r[0] for k in ("join_dir","k_score_long","k_many_sort","score_many")):
        print(r[0]); print("   "+"  ".join(f"{h}={v}" for h,v in zip(rows[0][1:],r[1:])))
PY
du -sh $O
