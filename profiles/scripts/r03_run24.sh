cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r3x; O=gpurun_out/r3x
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > $O/pytest.log 2>&1; tail -3 $O/pytest.log
timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu --cpu-targets 16e6 > $O/bench.json 2> $O/bench.log; grep "stage ms" $O/bench.log
python - <<'PY'
import json
j=json.load(open("gpurun_out/r3x/bench.json")); k=j["kernel_ms"]
print(round(j["ms_per_step"],1), {x:round(k[x]["ms"],2) for x in k if k[x]["ms"]>0}, (j.get("parity_full_index") or {}).get("mismatches"), (j.get("parity_sample") or {}).get("mismatches"))
PY
