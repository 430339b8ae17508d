#!/usr/bin/env python3
"""Re-test of the round-1 observation (world_size 1 only): variable-split all_to_all_single beyond ~1 GiB returned corrupt data
with torch 2.10 + RCCL 2.26.  Two ranks or more (torch.distributed.run): every rank sends `gib` GiB of a known pattern to every
peer, (a) through metabuli_amd.parallel._exchange (count all-gather + point-to-point rounds of at most 512 MiB), (b) through one
all_to_all_single call with split sizes; both are verified element by element."""
import os, sys
import torch
import torch.distributed as dist
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from metabuli_amd import parallel

rank = int(os.environ["RANK"]); world = int(os.environ["WORLD_SIZE"]); local = int(os.environ.get("LOCAL_RANK", rank))
torch.cuda.set_device(local); dev = torch.device("cuda", local)
dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
for gib in (0.5, 1.5, 3.0):
    rows = int(gib * 2**30) // 16                       # [rows, 2] int64 per peer
    counts = [rows + 1000 * ((rank + d) % 3) for d in range(world)]          # uneven splits
    send = torch.cat([torch.arange(c, device=dev, dtype=torch.int64).mul_(world * world).add_(rank * world + d).view(-1, 1).expand(-1, 2).contiguous() for d, c in enumerate(counts)])
    recv, rc = parallel._exchange(torch, dist, send, counts, 2, dev)
    ok_a, o = True, 0
    for src, c in enumerate(rc):
        exp = torch.arange(c, device=dev, dtype=torch.int64).mul_(world * world).add_(src * world + rank)
        ok_a = ok_a and bool((recv[o:o + c, 0] == exp).all()) and bool((recv[o:o + c, 1] == exp).all()); o += c
    out = torch.empty((sum(rc), 2), dtype=torch.int64, device=dev)
    dist.all_to_all_single(out, send, output_split_sizes=rc, input_split_sizes=counts)
    torch.cuda.synchronize()
    ok_b = bool((out == recv).all())
    flags = torch.tensor([int(ok_a), int(ok_b)], device=dev); dist.all_reduce(flags, op=dist.ReduceOp.MIN)
    if rank == 0:
        print(f"{gib} GiB per peer x {world} ranks: point-to-point rounds {'intact' if flags[0] else 'CORRUPT'}, one all_to_all_single {'intact' if flags[1] else 'CORRUPT'}", flush=True)
    del send, recv, out
dist.destroy_process_group()
