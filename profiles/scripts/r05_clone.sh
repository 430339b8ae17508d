# Round 5: the driver with three engines on ONE device (--devices 0,0,0): the resident index is read and decoded once and cloned to the two other engines CONCURRENTLY
# (mtb_index_clone from two host threads; device-to-device copies inside one GPU here, peer copies over xGMI on a node)
TAG=${1:-r05_clone}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; O=$R/gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
E2E_REPS=1 E2E_VARIANTS="|--devices 0,0,0" timeout 600 python profiles/scripts/e2e_big.py 4e9 20e6 64 4000000 > $O/${TAG}_4G_3engines.txt 2>&1; grep "cloned\|mtb_classify: 2\|max-reads\|database" $O/${TAG}_4G_3engines.txt | cut -c1-420
