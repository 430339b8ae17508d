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
