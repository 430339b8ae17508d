cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r3w; O=gpurun_out/r3w; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > $O/pytest.log 2>&1; tail -3 $O/pytest.log
timeout 600 python bench.py --cpu-reads 20000 --cpu-targets 16e6 --steps 2 --warmup 2 --seq-mode 3 --reads 200000 --read-len 10000 > $O/r03_final_bench_long.json 2> $O/bench_long.log; grep "stage ms" $O/bench_long.log; grep "parity" $O/bench_long.log | cut -c1-140
rm -rf $O/prof_long && mkdir -p $O/prof_long
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $R/$O/prof_long -o ks -- python $R/bench.py --steps 3 --warmup 2 --no-parity --seq-mode 3 --reads 200000 --read-len 10000 > $R/$O/prof_long/bench.json 2> $R/$O/prof_long/bench.log )
python profiles/scripts/rocpd_summary.py $(find $O/prof_long -name "*.db" | head -1) > $O/r03_final_long_rocprofv3_kernel_stats.txt 2>&1; head -8 $O/r03_final_long_rocprofv3_kernel_stats.txt | cut -c1-150
find $O/prof_long -name "*.db" -size +30M -delete
python - <<'PY'
import json
j=json.load(open("gpurun_out/r3w/r03_final_bench_long.json")); k=j["kernel_ms"]
print(round(j["ms_per_step"],1), round(j["value"],3), j["config"].get("gbp_per_s"), {x:round(k[x]["ms"],2) for x in k if k[x]["ms"]>0}, (j.get("parity_full_index") or {}).get("mismatches"), (j.get("parity_sample") or {}).get("mismatches"))
PY
