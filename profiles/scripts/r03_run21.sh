cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r3u; O=gpurun_out/r3u
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "long or fused or score or large" > $O/pytest.log 2>&1; tail -3 $O/pytest.log
timeout 600 python bench.py --cpu-reads 20000 --cpu-targets 16e6 --steps 2 --warmup 2 --seq-mode 3 --reads 200000 --read-len 10000 > $O/bench_long.json 2> $O/bench_long.log; grep "stage ms" $O/bench_long.log; grep "parity" $O/bench_long.log | cut -c1-140
MTB_LIB=$PWD/metabuli_amd/csrc/libmtb_xlprof.so timeout 400 python bench.py --no-cpu --no-parity --steps 1 --warmup 1 --seq-mode 3 --reads 200000 --read-len 10000 > $O/bench_long_prof.json 2> $O/bench_long_prof.log; grep "k_score_long phases" $O/bench_long_prof.log | tail -1
python - <<'PY'
import json
j=json.load(open("gpurun_out/r3u/bench_long.json")); k=j["kernel_ms"]
print(round(j["ms_per_step"],1), {x:round(k[x]["ms"],2) for x in k if k[x]["ms"]>0}, (j.get("parity_full_index") or {}).get("mismatches"), (j.get("parity_sample") or {}).get("mismatches"), j["config"].get("reads_scored_by_generic_kernel"))
PY
