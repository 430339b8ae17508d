#!/bin/bash
# One command for the first multi-GPU lease: the 1/2/4/8-GPU curve of both decompositions + the RCCL > 1 GiB re-test.
#   bash profiles/scripts/scale_all.sh [max_gpus]        -> gpurun_out/scale/{replicated,partitioned}_N.json, rccl_big_exchange.log
# Replicated: index replicated, reads sharded, no data-path collective (bench.py's default).
# Partitioned (row 2): every rank owns one value range of the index; two exchanges per batch over RCCL point-to-point.
set -x
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
MAXG=${1:-8}
O=gpurun_out/scale; mkdir -p $O
PORT=29650
for N in 1 2 4 8; do
  [ $N -gt $MAXG ] && break
  if [ $N -eq 1 ]; then
    timeout 600 python bench.py --gpus 1 --steps 5 --warmup 2 --no-cpu > $O/replicated_$N.json 2> $O/replicated_$N.log
    MTB_PART_TIMING=1 timeout 600 python bench.py --partitioned --species 24 --reads 2000000 --targets 2e9 --steps 5 --warmup 2 --no-parity > $O/partitioned_$N.json 2> $O/partitioned_$N.log
  else
    timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port $PORT bench.py --gpus $N --steps 5 --warmup 2 --no-cpu > $O/replicated_$N.json 2> $O/replicated_$N.log
    PORT=$((PORT+1))
    # every rank brings 2 M reads (weak on the reads); the 2 G-target index is cut into N value ranges, one per rank
    MTB_PART_TIMING=1 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port $PORT bench.py --gpus $N --partitioned --species 24 --reads 2000000 --targets 2e9 --steps 5 --warmup 2 --no-parity > $O/partitioned_$N.json 2> $O/partitioned_$N.log
    PORT=$((PORT+1))
  fi
  tail -1 $O/replicated_$N.json | cut -c1-300; tail -1 $O/partitioned_$N.json | cut -c1-300
done
# torch 2.10 + RCCL 2.26 returned corrupt data for variable-split all_to_all_single calls beyond ~1 GiB at world_size 1 (round 1);
# re-test at world_size >= 2 with the library's own exchange (point-to-point rounds) and with all_to_all_single
if [ $MAXG -ge 2 ]; then
  timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port $PORT profiles/scripts/rccl_big_exchange.py > $O/rccl_big_exchange.log 2>&1
  tail -5 $O/rccl_big_exchange.log
fi
