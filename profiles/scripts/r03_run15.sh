cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r3o; O=gpurun_out/r3o
timeout 600 python -m pytest tests/test_gpu_driver.py -m gpu -x -q > $O/pytest_driver.log 2>&1; tail -3 $O/pytest_driver.log
timeout 500 python profiles/scripts/e2e_driver.py 60e6 200e6 64 2000000,10000000 1,2 > $O/e2e.txt 2>&1; grep -v "^mtb_classify_batch_packed" $O/e2e.txt | grep "mtb_classify:\|run " | cut -c1-400
