cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r3r; O=gpurun_out/r3r
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > $O/pytest.log 2>&1; tail -5 $O/pytest.log
timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu --cpu-targets 16e6 > $O/bench.json 2> $O/bench.log; grep "stage ms" $O/bench.log; tail -3 $O/bench.log | cut -c1-300
MTB_NO_REC12=1 timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu --no-parity > $O/bench_rec16.json 2> $O/bench_rec16.log; grep "stage ms" $O/bench_rec16.log
timeout 400 python bench.py --cpu-reads 1000000 --cpu-targets 16e6 --steps 3 --warmup 2 --seq-mode 2 --reads 12500000 > $O/bench_paired.json 2> $O/bench_paired.log; grep "stage ms" $O/bench_paired.log; tail -2 $O/bench_paired.log | cut -c1-300
python - <<'PY'
import json
for f in ("bench","bench_rec16","bench_paired"):
    try:
        j=json.load(open(f"gpurun_out/r3r/{f}.json")); k=j["kernel_ms"]
        print(f, round(j["ms_per_step"],1), {x:round(k[x]["ms"],2) for x in ("score","score_fast","join","radix_scatter","radix_hist","extract_emit")}, (j.get("parity_full_index") or {}).get("mismatches"), (j.get("parity_sample") or {}).get("mismatches"))
    except Exception as e: print(f, "ERR", e)
PY
