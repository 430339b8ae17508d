cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r3l; O=gpurun_out/r3l
hipcc --offload-arch=gfx950 -O3 -o /tmp/swp profiles/scripts/store_window_probe.hip && timeout 120 /tmp/swp > $O/store_window_probe.txt 2>&1; cat $O/store_window_probe.txt
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > $O/pytest.log 2>&1; tail -3 $O/pytest.log
