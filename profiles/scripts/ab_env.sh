# A/B of environment knobs on one box: bash profiles/scripts/ab_env.sh "" "MTB_SORT_WIDE=1" ...   (each argument: env assignments, may be empty)
i=0
for rep in 1 2; do
for e in "$@"; do
tag=$(echo "${e:-default}" | tr ' =' '__')
env $e python bench.py --steps 5 --warmup 3 --no-parity $BENCH_FLAGS 2>&1 | tail -1 > gpurun_out/abenv_${tag}_$rep.json
done; done
python - <<PY
import json,glob
for f in sorted(glob.glob("gpurun_out/abenv_*.json")):
    try:
        d=json.loads(open(f).read()); print(f, round(d["ms_per_step"],1), {k:v["ms"] for k,v in d["roofline_all"].items()})
    except Exception as e: print(f, "failed:", open(f).read()[-400:])
PY
