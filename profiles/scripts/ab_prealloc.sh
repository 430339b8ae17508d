for rep in 1 2 3; do
python bench.py --steps 5 --warmup 3 --no-parity 2>&1 | tail -1 > gpurun_out/ab_plain_$rep.json
python bench.py --steps 5 --warmup 3 --no-parity --prealloc 2>&1 | tail -1 > gpurun_out/ab_prealloc_$rep.json
done
python - <<PY
import json,glob
for f in sorted(glob.glob("gpurun_out/ab_p*.json")):
    try:
        d=json.loads(open(f).read()); print(f, round(d["ms_per_step"],1), {k:v["ms"] for k,v in d["roofline_all"].items()})
    except Exception as e: print(f, "failed", open(f).read()[-300:])
PY
