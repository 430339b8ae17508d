# Round 5: end to end on a heavy-tailed database FROM FILES (2400 genomes, conserved segments, shared-run extras; 8 G targets), 100 M reads, driver defaults vs --async-results 1
TAG=${1:-r05_e2e}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; O=$R/gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
E2E_WORLD=heavy E2E_REPS=2 E2E_VARIANTS="|--async-results 1" timeout 1200 python profiles/scripts/e2e_big.py 8e9 100e6 64 4000000,8000000 > $O/${TAG}_heavy_8G.txt 2>&1; grep "heavy\|database\|mtb_classify: 1\|max-reads\|working" $O/${TAG}_heavy_8G.txt | cut -c1-460
