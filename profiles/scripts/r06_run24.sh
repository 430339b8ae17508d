#!/bin/bash
# round 6, GPU call 24: SQ counters of the final kernels -- headline (window form pinned) and 10 M reads of held-out genomes; three --pmc passes each, kernel trace only,
# every pass summarised (last dispatch of every kernel) before its csv is deleted (call 22 deleted them first)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=$R/gpurun_out/r06_run24; mkdir -p $O; S=/tmp/mtb_sq; rm -rf $S; mkdir -p $S; export TMPDIR=/tmp
pass() { tag=$1; d=$2; c="$3"; shift 3
  ( cd /tmp && timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $S/${tag}_$d -- python $R/bench.py --steps 1 --warmup 1 --no-parity --no-legs --no-cpu "$@" > $O/${tag}_$d.log 2>&1 ); echo "$tag pass $d rc=$?"
  python profiles/scripts/pmc_summary.py $S/${tag}_$d > $O/${tag}_$d.tsv 2>> $O/summary.err; rm -rf $S/${tag}_$d; }
A="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VALU"
B="SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA"
C="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_FLAT SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE GRBM_COUNT"
export MTB_JOIN_VARIANT=window
pass headline a "$A"; pass headline b "$B"; pass headline c "$C"
pass heldout a "$A" --reads-from heldout; pass heldout b "$B" --reads-from heldout; pass heldout c "$C" --reads-from heldout
wc -c $O/*.tsv
