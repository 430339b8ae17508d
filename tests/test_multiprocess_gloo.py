"""N>1 path on CPU: two processes (gloo) shard one read batch with
metabuli_amd.parallel exactly as the GPU ranks do, classify their shards (the
oracle stands in for the per-rank engine here), and the gathered per-read
results and the all-reduced per-taxon counts must equal the single-process run."""
import os
import sys

import numpy as np
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, dbdir, npz, out):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist
    from helpers import Oracle, default_params
    from metabuli_amd import parallel
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = np.load(npz)
    orc = Oracle()
    p = default_params(seq_mode=1, syncmer=1)
    tax = orc.load_taxonomy(os.path.join(dbdir, "taxonomy"))
    db = orc.open_db(dbdir, tax, p)
    b, o, lo, hi = parallel.shard_reads(g["bases"], g["offs"], rank, world)
    R = orc.classify(db, tax, p, b, o)
    mx = orc.lib.orc_tax_max_id(tax)
    counts = parallel.allreduce_tax_counts(parallel.tax_counts(R["results"]["classification"], mx), dist)
    gathered = [None] * world
    dist.all_gather_object(gathered, (lo, hi, R["results"]["classification"].tolist(), R["results"]["score"].tolist()))
    dist.barrier()
    if rank == 0:
        cls = np.zeros(len(g["offs"]) - 1, np.int32); sc = np.zeros(len(g["offs"]) - 1, np.float32)
        for lo_, hi_, c, s in gathered:
            cls[lo_:hi_] = c; sc[lo_:hi_] = s
        np.savez(out, cls=cls, score=sc, counts=counts)
    dist.destroy_process_group()


def test_two_rank_sharding_matches_single_process(orc, tmp_path):
    from conftest import Toy
    from metabuli_amd import parallel
    t = Toy(orc, tmp_path / "db", syncmer=1, paired=False, seed=12, n_reads=101)
    npz = str(tmp_path / "reads.npz"); out = str(tmp_path / "out.npz")
    np.savez(npz, bases=t.b1, offs=t.o1)
    mp.spawn(_worker, args=(2, 29541 + os.getpid() % 500, t.dbdir, npz, out), nprocs=2, join=True)
    r = np.load(out)
    ro = t.ref["results"]
    assert (r["cls"] == ro["classification"]).all()
    assert (r["score"].view(np.uint32) == ro["score"].view(np.uint32)).all()
    mx = orc.lib.orc_tax_max_id(t.tax)
    assert (r["counts"] == parallel.tax_counts(ro["classification"], mx)).all()
    assert int(r["counts"].sum()) == 101


def test_shard_range_partition():
    from metabuli_amd.parallel import shard_range
    for n in (0, 1, 7, 8, 9, 1000):
        for w in (1, 2, 3, 8):
            spans = [shard_range(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))
            assert max(b - a for a, b in spans) - min(b - a for a, b in spans) <= 1


def test_bench_json_line_is_the_last_line_of_the_combined_stdout(tmp_path):
    """bench.py under torch.distributed.run: every rank shares one stdout, and C libraries (RCCL prints its path through C stdio,
    block-buffered on a pipe) flush at process exit -- after a JSON line printed earlier.  bench.silence_other_ranks + bench.finish
    must leave rank 0's JSON line as the LAST line and the only JSON line.  Two gloo ranks, a C-level printf on every rank."""
    import json
    import subprocess
    script = tmp_path / "two_ranks.py"
    script.write_text(f"""
import ctypes, os, sys
sys.path.insert(0, {ROOT!r})
import torch.distributed as dist
import bench
rank = int(os.environ["RANK"])
bench.silence_other_ranks(rank)
dist.init_process_group("gloo", rank=rank, world_size=int(os.environ["WORLD_SIZE"]))
libc = ctypes.CDLL(None)
libc.printf(b"C-level chatter of rank %d (buffered until exit on a pipe)\\n", rank)
print("python-level chatter of rank", rank, flush=True)
bench.finish(dist, '{{"value": 1.5, "n_gpus": 2}}' if rank == 0 else None)
""")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    port = 29600 + os.getpid() % 300
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), str(script)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert json.loads(lines[-1]) == {"value": 1.5, "n_gpus": 2}
    assert sum(1 for l in lines if l.lstrip().startswith("{")) == 1
    assert not any("rank 1" in l for l in lines)              # nothing of the other rank reaches the shared stdout
    assert any("C-level chatter of rank 0" in l for l in lines[:-1])
