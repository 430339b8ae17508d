#!/bin/bash
# SQ counter passes of the default bench command (rocprofv3 --pmc, own runs, kernel trace only): instruction mix, issue /
# wait split, LDS conflicts per kernel.  Usage on the GPU box: bash profiles/scripts/pmc_sq.sh <tag> [bench args...]
# Output: gpurun_out/pmc_<tag>_{a,b,c}/ (csv).  Summaries are made by profiles/scripts/pmc_summary.py.
tag=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $R/gpurun_out/counters_list.txt 2>&1
run() { d=$1; shift; c="$1"; shift
  timeout 900 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/pmc_${tag}_$d -- python $R/bench.py --steps 1 --warmup 1 --no-parity "$@" > $R/gpurun_out/pmc_${tag}_$d.log 2>&1
  echo "pass $d rc=$?"; }
run a "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VALU" "$@"
run b "SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA" "$@"
run c "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_FLAT SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE GRBM_COUNT" "$@"
if [ -z "$MTB_PMC_NO_TCC" ]; then
run d "FETCH_SIZE" "$@"
run e "WRITE_SIZE" "$@"
fi
find $R/gpurun_out -name "*counter_collection.csv" -size +60M -delete
ls -la $R/gpurun_out/pmc_${tag}_* | head -40
