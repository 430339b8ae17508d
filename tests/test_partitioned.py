"""Range-partitioned index: every rank owns one value range of the target array, query
metamers travel to the owner (all-to-all #1), matches travel home (all-to-all #2).

CPU (gloo, world_size 2): metabuli_amd.parallel.classify_partitioned drives the exchange exactly as on GPUs; the
per-rank stages are played by the oracle / the host build of the kernel arithmetic, and the gathered per-read
results must equal the single-process oracle run on the whole index.
GPU (-m gpu): the same with libmtb's mtb_part_* stage calls, two processes sharing cuda:0, ranges loaded from the
database files through the split checkpoints (mtb_index_open_part)."""
import os
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
AAMASK = np.uint64(0xFFFFFFFFFF000000)


class EmuStages:
    """CPU stand-in for metabuli_amd.parallel.GpuStages (tests only)."""

    def __init__(self, orc, emu, p, db, tax, values, info, t2s, lo, hi, is_last, bases, offs):
        import torch
        self.torch, self.orc, self.emu, self.p, self.db, self.tax = torch, orc, emu, p, db, tax
        a, b = np.searchsorted(values, [lo, hi]) if hi != np.uint64(2**64 - 1) else (np.searchsorted(values, lo), len(values))
        self.values = values[a:b].copy(); self.info = info[a:b].copy(); self.t2s = t2s
        if not is_last:           # the "last entry is never a candidate" rule belongs to the last range only
            self.values = np.append(self.values, np.uint64(2**64 - 1)); self.info = np.append(self.info, np.uint32(0))
        self.bases, self.offs = bases, offs

    def extract_sorted(self, bounds):
        k, self.ql, self.ql2 = self.orc.extract_batch(self.p, self.bases, self.offs)
        k = self.orc.sort_kmers(k)
        b = np.asarray(bounds, np.uint64)
        kt = self.torch.from_numpy(k.view(np.int64).reshape(-1, 2).copy())
        if os.environ.get("MTB_TEST_OVERLAP"):
            # what libmtb's product path does: runs cut at the granularity of the 30-bit amino-acid prefix, so a run also carries the
            # metamers that share their prefix with its upper bound (the neighbour gets them too and finds nothing for the foreign ones)
            pre = b >> np.uint64(34)
            lo = np.searchsorted(k["value"], pre << np.uint64(34)); lo[0] = 0
            hi = np.append(np.searchsorted(k["value"], (pre[1:] + np.uint64(1)) << np.uint64(34)), len(k))
            return kt, [int(max(0, h - l)) for l, h in zip(lo, hi)], [int(x) for x in lo]
        pos = np.searchsorted(k["value"], b)
        counts = np.diff(np.append(pos, len(k)))
        return kt, [int(c) for c in counts]

    def join(self, run):
        from helpers import kmer_dt
        q = run.numpy().copy().view(np.uint64).reshape(-1).view(kmer_dt)
        m = self.emu.join(self.values, self.info, self.t2s, 0xFFFFFFFF, self.p.kmer_format, q)
        return self.torch.from_numpy(m.view(np.int64).reshape(-1, 3).copy())

    def score(self, matches):
        from helpers import match_dt
        m = matches.numpy().copy().view(np.uint64).reshape(-1).view(match_dt)
        m = self.orc.sort_matches(m)
        return self.orc.score(self.db, self.tax, self.p, m, len(self.offs) - 1, self.ql, self.ql2)


def _cpu_worker(rank, world, port, dbdir, npz, out):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist
    from helpers import Oracle, Emu, default_params
    import metabuli_amd
    from metabuli_amd import parallel
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    if os.environ.get("MTB_TEST_XCHG"):          # force the multi-round (chunked) exchange
        parallel._XCHG_BYTES = int(os.environ["MTB_TEST_XCHG"])
    g = np.load(npz)
    orc, emu = Oracle(), Emu()
    p = default_params(seq_mode=1, syncmer=1)
    tax = orc.load_taxonomy(os.path.join(dbdir, "taxonomy"))
    db = orc.open_db(dbdir, tax, p)
    bounds = metabuli_amd.part_bounds(dbdir, world)
    _, t2s = emu.load_taxonomy(os.path.join(dbdir, "taxonomy"), g["taxids"])
    b, o, lo, hi = parallel.shard_reads(g["bases"], g["offs"], rank, world)
    hi_b = bounds[rank + 1] if rank + 1 < world else np.uint64(2**64 - 1)
    st = EmuStages(orc, emu, p, db, tax, g["values"], g["taxids"].astype(np.uint32), t2s, bounds[rank], hi_b, rank == world - 1, b, o)
    res, tt, tc = parallel.classify_partitioned(st, bounds, dist)
    gathered = [None] * world
    dist.all_gather_object(gathered, (lo, hi, res["classification"].tolist(), res["score"].tolist(), tt.tolist(), tc.tolist()))
    if rank == 0:
        n = len(g["offs"]) - 1
        cls = np.zeros(n, np.int32); sc = np.zeros(n, np.float32); att, atc = [], []
        for lo_, hi_, c, s, t1, t2 in gathered:
            cls[lo_:hi_] = c; sc[lo_:hi_] = s; att += t1; atc += t2
        np.savez(out, cls=cls, score=sc, tt=np.array(att, np.int32), tc=np.array(atc, np.uint32))
    dist.barrier()
    dist.destroy_process_group()


def _check(out, ref):
    r = np.load(out)
    ro = ref["results"]
    assert (r["cls"] == ro["classification"]).all()
    assert (r["score"].view(np.uint32) == ro["score"].view(np.uint32)).all()
    assert (r["tt"] == ref["tc_tax"]).all() and (r["tc"] == ref["tc_cnt"]).all()


def test_part_bounds_are_amino_acid_boundaries(orc, tmp_path):
    import metabuli_amd
    from conftest import Toy
    t = Toy(orc, tmp_path / "db", syncmer=1, paired=False, seed=21, n_reads=4)
    for world in (1, 2, 3, 8):
        b = metabuli_amd.part_bounds(t.dbdir, world)
        assert b[0] == 0 and (np.diff(b.astype(np.float64)) >= 0).all()
        assert ((b & ~AAMASK) == 0).all()
        cuts = np.searchsorted(t.values, b)
        sizes = np.diff(np.append(cuts, len(t.values)))
        assert sizes.sum() == len(t.values)
        if world > 1:        # balanced within a few checkpoints, and no amino-acid group straddles a cut
            assert sizes.max() - sizes.min() < len(t.values) // 50 + 200
            for c in cuts[1:]:
                assert (t.values[c - 1] & AAMASK) != (t.values[c] & AAMASK)


@pytest.mark.parametrize("xchg_bytes,overlap", [(None, False), (20000, False), (None, True), (20000, True)])
def test_two_rank_partitioned_index_matches_single_process(orc, tmp_path, xchg_bytes, overlap, monkeypatch):
    from conftest import Toy
    t = Toy(orc, tmp_path / "db", syncmer=1, paired=False, seed=22, n_reads=90)
    npz = str(tmp_path / "in.npz"); out = str(tmp_path / "out.npz")
    np.savez(npz, bases=t.b1, offs=t.o1, values=t.values, taxids=t.taxids)
    if xchg_bytes:
        monkeypatch.setenv("MTB_TEST_XCHG", str(xchg_bytes))
    if overlap:
        monkeypatch.setenv("MTB_TEST_OVERLAP", "1")
    mp.spawn(_cpu_worker, args=(2, 30100 + os.getpid() % 500, t.dbdir, npz, out), nprocs=2, join=True)
    _check(out, t.ref)


# ------------------------------------------------------------------ GPU
def _gpu_worker(rank, world, port, dbdir, npz, out, seq_mode):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import torch.distributed as dist
    import metabuli_amd as M
    from metabuli_amd import parallel
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)     # one GPU on the test box: payload staged through the host
    g = np.load(npz)
    ctx = M.Context(0)
    p = M.default_params(seq_mode=seq_mode, syncmer=1)
    bounds = M.part_bounds(dbdir, world)
    ix = ctx.open_index_part(dbdir, p, rank, world)
    b, o, lo, hi = parallel.shard_reads(g["bases"], g["offs"], rank, world)
    dev = torch.device("cuda:0")
    Stages = parallel.GpuStages
    if os.environ.get("MTB_HIPEMU"):
        # the emulated build (tests/hipemu): "device" memory is host memory -- CPU tensors, nothing to fence, the context-owned buffer seen through numpy
        dev = torch.device("cpu")

        class Stages(parallel.GpuStages):
            def _fence(self):
                pass

            def extract_sorted(self, bounds):
                import ctypes as C
                b, o, b2, o2 = self.reads
                ptr, nk, counts, starts = self.ctx.part_extract(self.params, b.data_ptr(), o.data_ptr(), b2.data_ptr() if b2 is not None else 0,
                                                                o2.data_ptr() if o2 is not None else 0, self.n_reads, bounds, overlapping=self.overlapping)
                if nk == 0:
                    return torch.empty((0, 2), dtype=torch.int64), [0] * len(counts), [0] * len(counts)
                view = np.ctypeslib.as_array((C.c_int64 * (2 * int(nk))).from_address(int(ptr))).reshape(int(nk), 2)
                return torch.from_numpy(view), [int(c) for c in counts], [int(x) for x in starts]
    st = Stages(ctx, ix, p, dev)
    st.overlapping = not os.environ.get("MTB_TEST_PART_LEGACY")
    if seq_mode == 2:
        b2, o2, _, _ = parallel.shard_reads(g["bases2"], g["offs2"], rank, world)
        st.set_reads(torch.from_numpy(b.copy()).to(dev), torch.from_numpy(o.astype(np.int64)).to(dev), hi - lo,
                     torch.from_numpy(b2.copy()).to(dev), torch.from_numpy(o2.astype(np.int64)).to(dev))
    else:
        st.set_reads(torch.from_numpy(b.copy()).to(dev), torch.from_numpy(o.astype(np.int64)).to(dev), hi - lo)
    res, tt, tc = parallel.classify_partitioned(st, bounds, dist)
    slot_reads = int(ctx.last_stats().n_slot_reads) if st.overlapping else -1
    gathered = [None] * world
    dist.all_gather_object(gathered, (lo, hi, res["classification"].tolist(), res["score"].tolist(), tt.tolist(), tc.tolist(), ix.num_targets, slot_reads))
    if rank == 0:
        n = len(g["offs"]) - 1
        cls = np.zeros(n, np.int32); sc = np.zeros(n, np.float32); att, atc = [], []; T = 0
        slot_ok = True
        for lo_, hi_, c, s, t1, t2, tn, sr in gathered:
            cls[lo_:hi_] = c; sc[lo_:hi_] = s; att += t1; atc += t2; T += tn
            slot_ok = slot_ok and (sr == -1 or sr == hi_ - lo_)      # the product path really took the slot segments
        np.savez(out, cls=cls, score=sc, tt=np.array(att, np.int32), tc=np.array(atc, np.uint32), T=T, slot_ok=slot_ok)
    dist.barrier()
    ix.close(); ctx.close()
    dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize("world,legacy", [(2, False), (3, False), (2, True)])
def test_gpu_partitioned_two_processes(orc, tmp_path, world, legacy, monkeypatch):
    """legacy = False: the product path (ordinal tags, prefix-granular overlapping runs, directory join at the owners, matches placed
    into the home rank's slot segments, slot scorers); True: consecutive runs, regroup + segment sort at home"""
    from conftest import Toy
    t = Toy(orc, tmp_path / "db", syncmer=1, paired=False, seed=23, n_reads=200)
    npz = str(tmp_path / "in.npz"); out = str(tmp_path / "out.npz")
    np.savez(npz, bases=t.b1, offs=t.o1)
    if legacy:
        monkeypatch.setenv("MTB_TEST_PART_LEGACY", "1")
    mp.spawn(_gpu_worker, args=(world, 30700 + os.getpid() % 500, t.dbdir, npz, out, 1), nprocs=world, join=True)
    _check(out, t.ref)
    assert int(np.load(out)["T"]) == len(t.values) and bool(np.load(out)["slot_ok"])


@pytest.mark.gpu
@pytest.mark.parametrize("seq_mode", [2, 3])
def test_gpu_partitioned_pairs_and_long_reads(orc, tmp_path, seq_mode):
    """the partitioned batch against the oracle for read pairs (slot path: mate offsets in the ordinal-tagged positions) and for long
    reads (exact-order runs, regroup + segment sort at home)"""
    from conftest import Toy
    if seq_mode == 2:
        t = Toy(orc, tmp_path / "db", syncmer=1, paired=True, seed=25, n_reads=200)
    else:
        t = Toy(orc, tmp_path / "db", syncmer=1, paired=False, seed=26, n_reads=16, length=3000, seq_mode=3, err=0.05, lognormal=True)
    npz = str(tmp_path / "in.npz"); out = str(tmp_path / "out.npz")
    if seq_mode == 2:
        np.savez(npz, bases=t.b1, offs=t.o1, bases2=t.b2, offs2=t.o2)
    else:
        np.savez(npz, bases=t.b1, offs=t.o1)
    mp.spawn(_gpu_worker, args=(2, 30900 + os.getpid() % 500, t.dbdir, npz, out, seq_mode), nprocs=2, join=True)
    r = np.load(out); ro = t.ref["results"]; amb = ro["flag"] != 0
    assert ((r["cls"] == ro["classification"]) | amb).all()
    assert ((r["score"].view(np.uint32) == ro["score"].view(np.uint32)) | amb).all()
    if not amb.any():
        assert (r["tt"] == t.ref["tc_tax"]).all() and (r["tc"] == t.ref["tc_cnt"]).all()
    assert int(r["T"]) == len(t.values)
    if seq_mode == 2:
        assert bool(r["slot_ok"])


@pytest.mark.gpu
@pytest.mark.parametrize("no_dir", [False, True])
def test_gpu_partitioned_long_runs_and_the_bisection_fallback(orc, tmp_path, no_dir, monkeypatch):
    """(a) candidate runs of > 70 entries on the owner side: k_join_dir's dense-list mode scans them with the whole wave (count before
    the workgroup's reservation, ballot-ranked emission behind it); (b) ADVICE r3: an owner without a directory (MTB_NO_DIR) falls back
    to the bisection join, whose tile windows must follow the SENDER's sort granularity (slot-mode runs are ordered on bits [34, 64)
    only) -- no match may be lost, and the records (pad = 0) still reach the home rank's slots through the tails"""
    from conftest import HotToy
    t = HotToy(orc, tmp_path / "db", seq_mode=1, n_reads=200)
    npz = str(tmp_path / "in.npz"); out = str(tmp_path / "out.npz")
    np.savez(npz, bases=t.b1, offs=t.o1)
    if no_dir:
        monkeypatch.setenv("MTB_NO_DIR", "1")
    mp.spawn(_gpu_worker, args=(2, 31100 + os.getpid() % 500, t.dbdir, npz, out, 1), nprocs=2, join=True)
    _check(out, t.ref)
    assert int(np.load(out)["T"]) == len(t.values) and bool(np.load(out)["slot_ok"])


@pytest.mark.gpu
def test_gpu_partitions_concatenate_to_the_index(orc, tmp_path):
    import metabuli_amd as M
    from conftest import Toy
    t = Toy(orc, tmp_path / "db", syncmer=1, paired=False, seed=24, n_reads=4)
    ctx = M.Context(0)
    for world in (1, 2, 5):
        vs, infos = [], []
        for r in range(world):
            p = M.default_params(seq_mode=1, syncmer=1)
            ix = ctx.open_index_part(t.dbdir, p, r, world)
            v, i = ix.download(); vs.append(v); infos.append(i); ix.close()
        assert (np.concatenate(vs) == t.values).all()
        assert (np.concatenate(infos) == t.taxids.astype(np.uint32)).all()
    # device views of a resident index cut the same way
    p = M.default_params(seq_mode=1, syncmer=1)
    full = ctx.open_index(t.dbdir, p)
    b = M.part_bounds(t.dbdir, 3)
    parts = [full.slice(b[r], b[r + 1] if r < 2 else 2**64 - 1, r == 2) for r in range(3)]
    assert (np.concatenate([q.download()[0] for q in parts]) == t.values).all()
    for q in parts:
        q.close()
    full.close(); ctx.close()


def test_part_bounds_of_tiny_and_coarse_databases(orc, tmp_path):
    """fewer usable split checkpoints than ranges: the surplus ranges are empty (their bound repeats the next one or is
    the maximum), never overlapping"""
    import metabuli_amd
    from helpers import default_params
    from metabuli_amd import synth
    p = default_params(seq_mode=1, syncmer=1)
    w = synth.make_world(seed=5, n_genera=1, species_per_genus=1, strains_per_species=1, genome_len=2000, with_euk=False)
    g = w.genomes[0][1]
    k, _, _ = orc.extract_batch(default_params(seq_mode=3, syncmer=1), g, np.array([0, len(g)], np.uint64))
    vals = np.unique(k["value"]); tids = np.full(len(vals), w.genomes[0][0], np.int32)
    # (a) default 4096 checkpoints on ~1.5 k metamers: no checkpoint is ever set
    d = str(tmp_path / "tiny"); os.makedirs(d)
    w.tax.write(os.path.join(d, "taxonomy")); orc.write_db(d, vals, tids, p)
    b = metabuli_amd.part_bounds(d, 4)
    assert b[0] == 0 and (np.diff(b.astype(np.float64)) >= 0).all()
    sizes = np.diff(np.append(np.searchsorted(vals, b), len(vals)))
    assert (sizes > 0).sum() == 1 and sizes.sum() == len(vals)          # one range holds everything, the others are empty
    # (b) 6 checkpoints, 16 ranges
    d2 = str(tmp_path / "coarse"); os.makedirs(d2)
    w.tax.write(os.path.join(d2, "taxonomy")); orc.write_db(d2, vals, tids, p, split_num=6)
    b = metabuli_amd.part_bounds(d2, 16)
    assert b[0] == 0 and (np.diff(b.astype(np.float64)) >= 0).all() and ((b & ~AAMASK) == 0).all()
    cuts = np.searchsorted(vals, b)
    sizes = np.diff(np.append(cuts, len(vals)))
    assert sizes.sum() == len(vals) and (sizes > 0).sum() <= 6
    for c in cuts[(cuts > 0) & (cuts < len(vals))]:
        assert (vals[c - 1] & AAMASK) != (vals[c] & AAMASK)


@pytest.mark.gpu
def test_gpu_partitions_with_empty_ranges(orc, tmp_path):
    """more ranges than checkpoints: empty ranges open as empty indices, the others still concatenate to the database"""
    import metabuli_amd as M
    from helpers import default_params
    from metabuli_amd import synth
    p = default_params(seq_mode=1, syncmer=1)
    w = synth.make_world(seed=6, n_genera=1, species_per_genus=2, strains_per_species=1, genome_len=3000, with_euk=False)
    d = str(tmp_path / "db"); os.makedirs(d)
    from helpers import build_toy_db
    vals, tids = build_toy_db(orc, w, p, d)
    orc.write_db(d, vals, tids, p, split_num=6)
    ctx = M.Context(0)
    for world in (4, 16):
        vs, n_empty = [], 0
        for r in range(world):
            mp = M.default_params(seq_mode=1, syncmer=1)
            ix = ctx.open_index_part(d, mp, r, world)
            v, i = ix.download(); vs.append(v); n_empty += len(v) == 0; ix.close()
        assert (np.concatenate(vs) == vals).all()
        assert n_empty >= world - 6
    ctx.close()
