TAG=${1:-r05_suite}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; O=$R/gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q --maxfail=10 -p no:cacheprovider > $O/${TAG}_pytest_gpu.log 2>&1; tail -n 15 $O/${TAG}_pytest_gpu.log | cut -c1-300
