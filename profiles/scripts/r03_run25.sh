# final pass of round 3 on the final code: GPU suite, smoke, FETCH / WRITE passes, headline bench, kernel stats
TAG=r03_final
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; O=gpurun_out; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q > $O/${TAG}_pytest_gpu.log 2>&1; tail -3 $O/${TAG}_pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
( cd /tmp
  for pass in "d FETCH_SIZE" "e WRITE_SIZE"; do set -- $pass
    timeout 600 rocprofv3 --pmc $2 --kernel-trace --output-format csv -d $R/$O/pmc_${TAG}_$1 -- python $R/bench.py --steps 1 --warmup 1 --no-parity > $R/$O/pmc_${TAG}_$1.log 2>&1; echo "pmc pass $1 rc=$?"
  done )
python profiles/scripts/pmc_summary.py $O/pmc_${TAG}_d $O/pmc_${TAG}_e > $O/${TAG}_pmc_counters.tsv 2> $O/${TAG}_pmc_summary.err
python profiles/scripts/make_pmc_traffic.py $O/pmc_${TAG} 10000000 150 16024359718 1 "profiles/${TAG}_pmc_counters.tsv (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of bench.py --steps 1 --warmup 1 --no-parity)" > $O/${TAG}_pmc_traffic_print.json 2> $O/${TAG}_pmc_traffic.err; cp profiles/pmc_traffic.json $O/${TAG}_pmc_traffic.json
find $O -name "*counter_collection.csv" -size +20M -delete
timeout 900 python bench.py --steps 10 --warmup 3 > $O/${TAG}_bench.json 2> $O/${TAG}_bench.log; grep "stage ms" $O/${TAG}_bench.log
rm -rf $O/prof_ks && mkdir -p $O/prof_ks
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $R/$O/prof_ks -o ks -- python $R/bench.py --steps 5 --warmup 2 --no-parity > $R/$O/prof_ks/bench.json 2> $R/$O/prof_ks/bench.log )
python profiles/scripts/rocpd_summary.py $(find $O/prof_ks -name "*.db" | head -1) > $O/${TAG}_rocprofv3_kernel_stats.txt 2>&1; head -12 $O/${TAG}_rocprofv3_kernel_stats.txt | cut -c1-150
find $O/prof_ks -name "*.db" -size +30M -delete
python - <<'PY'
import json
j=json.load(open("gpurun_out/r03_final_bench.json")); k=j["kernel_ms"]
print(round(j["ms_per_step"],2), round(j["value"],2), {x:round(k[x]["ms"],2) for x in k if k[x]["ms"]>0}, (j.get("parity_full_index") or {}).get("mismatches"), (j.get("parity_sample") or {}).get("mismatches"), j["roofline"]["frac"], j["roofline"]["effective"], j["cpu_baseline"]["value"])
PY
