cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r3q; O=gpurun_out/r3q
run() { name=$1; shift; env "$@" timeout 200 python bench.py --steps 6 --warmup 2 --no-parity --no-cpu --streams $S > $O/$name.json 2> $O/$name.log; python -c "
import json; j=json.load(open('$O/$name.json')); print('$name', round(j['ms_per_step'],1), round(j['value'],1))"; }
S=1 run s1 MTB_X=0
S=2 run s2_lock MTB_X=0
S=2 run s2_stag25 MTB_LANE_STAGGER_MS=25
S=2 run s2_c2_stag12 MTB_LANE_STAGGER_MS=12 MTB_CHUNKS_PER_STREAM=2
S=2 run s2_c3_stag9 MTB_LANE_STAGGER_MS=9 MTB_CHUNKS_PER_STREAM=3
S=3 run s3_c2_stag8 MTB_LANE_STAGGER_MS=8 MTB_CHUNKS_PER_STREAM=2
