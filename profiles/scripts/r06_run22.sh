#!/bin/bash
# round 6, GPU call 22: SQ counters of the final kernels -- headline (window form pinned) and 10 M reads of held-out genomes (three --pmc passes each, kernel trace only)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=$R/gpurun_out/r06_run22; mkdir -p $O
export MTB_PMC_NO_TCC=1
MTB_JOIN_VARIANT=window bash profiles/scripts/pmc_sq.sh r06f_headline --no-legs --no-cpu > $O/sq_headline.log 2>&1
python profiles/scripts/pmc_summary.py gpurun_out/pmc_r06f_headline_a gpurun_out/pmc_r06f_headline_b gpurun_out/pmc_r06f_headline_c > $O/r06_final_pmc_sq_headline.txt 2> $O/sum1.err
bash profiles/scripts/pmc_sq.sh r06f_heldout --reads-from heldout --no-legs --no-cpu > $O/sq_heldout.log 2>&1
python profiles/scripts/pmc_summary.py gpurun_out/pmc_r06f_heldout_a gpurun_out/pmc_r06f_heldout_b gpurun_out/pmc_r06f_heldout_c > $O/r06_final_pmc_sq_heldout.txt 2> $O/sum2.err
rm -rf gpurun_out/pmc_r06f_*
grep -A1 "k_join_dir\|k_many_sort\|k_score_long\|k_score_many" $O/r06_final_pmc_sq_heldout.txt | cut -c1-900 | head -30
