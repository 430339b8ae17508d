TAG=${1:-r05_c23}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; O=$R/gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q --maxfail=5 -p no:cacheprovider -k "zeroed_device_memory or many or long or bench or target_windows or prefetched or hbm_budgeted or partitioned or driver_on_two or a_few_long_reads" > $O/${TAG}_pytest_subset.log 2>&1; tail -n 4 $O/${TAG}_pytest_subset.log | cut -c1-300
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
