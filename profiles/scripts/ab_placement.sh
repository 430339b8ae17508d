# the slot buffer's placement probe on / off over several processes: join ms per process
for rep in 1 2 3 4; do
for e in "MTB_PLACEMENT_VERBOSE=1" "MTB_NO_PLACEMENT_PROBE=1"; do
  env $e python bench.py --steps 4 --warmup 3 --no-parity 2> gpurun_out/place.err | tail -1 > gpurun_out/place.json
  python - "$e" <<PY
import json,sys
d=json.loads(open("gpurun_out/place.json").read())
probe=[l.strip() for l in open("gpurun_out/place.err") if "placement probe" in l]
print(sys.argv[1], "step %.1f" % d["ms_per_step"], "join %.1f" % d["roofline_all"]["join"]["ms"], probe[:2])
PY
done; done
