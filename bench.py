#!/usr/bin/env python3
"""bench.py -- `metabuli classify` hot path on MI355X: Mreads/s (+ Gbp/s) with
the reads and the target index already resident in HBM.

One "step" = one pass of the whole hot path (extract -> radix sort -> join
against the resident index -> regroup/segment sort -> per-read scoring) over one
batch of synthetic reads (BASELINE.json configs[1]: 10 M x 150 bp single-end
vs a GTDB-scale synthetic index).  One process per GPU; reads are sharded, the
index is replicated, there is no data-path collective (weak scaling).

    python bench.py --gpus 1 --steps 3 --warmup 1
    python -m torch.distributed.run --nproc-per-node 8 ... bench.py --gpus 8 ...

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def build_world(seed, n_species, genome_len, n_filler_species):
    from metabuli_amd import synth
    return synth.make_world(seed=seed, n_genera=max(1, n_species // 4), species_per_genus=4, strains_per_species=1,
                            genome_len=genome_len, n_filler_species=n_filler_species)


def build_world_fast(torch, dev, seed, n_species, genome_len, n_filler_species):
    """Same shape as synth.make_world (root -> {Bacteria, Eukaryota} -> genus -> 4 species -> 1 strain, genus divergence 15 %,
    strain divergence 1 %) for THOUSANDS of genomes: sequences are drawn and mutated on the device (Bernoulli substitutions) instead
    of numpy's choice-without-replacement per genome.  Used for the genome-diversity runs (--species >= 200)."""
    from metabuli_amd import synth
    g = torch.Generator(device=dev); g.manual_seed(seed)
    tax = synth.Taxonomy()
    tax.add(1, 1, "no rank", "root"); tax.add(2, 1, "superkingdom", "Bacteria"); tax.add(3, 1, "superkingdom", "Eukaryota")
    nxt = 4
    n_genera = max(1, n_species // 4)
    acgt = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=dev)

    def mutate(x, rate):
        m = torch.rand(x.shape, generator=g, device=dev) < rate
        sub = (x + torch.randint(1, 4, x.shape, generator=g, device=dev, dtype=torch.uint8)) & 3       # a DIFFERENT base
        return torch.where(m, sub, x)
    genomes, species = [], []
    for gi in range(n_genera):
        dom = 3 if gi == n_genera - 1 else 2
        gid = nxt; nxt += 1
        tax.add(gid, dom, "genus", f"Genus{gi}")
        anc = torch.randint(0, 4, (genome_len,), generator=g, device=dev, dtype=torch.uint8)
        for sidx in range(4):
            sid = nxt; nxt += 1
            tax.add(sid, gid, "species", f"Genus{gi} species{sidx}")
            species.append(sid)
            tid = nxt; nxt += 1
            tax.add(tid, sid, "no rank", f"Genus{gi} species{sidx} strain0")
            genomes.append((tid, acgt[mutate(mutate(anc, 0.15), 0.01).long()].cpu().numpy()))
    lo = nxt
    for i in range(n_filler_species):
        tax.add(nxt, 2, "species", f"filler{i}"); nxt += 1
    return synth.World(tax, genomes, species, lo, nxt - 1)


def extract_targets(ctx, M, world, params, torch=None, dev=None, genomes_per_call=48):
    """Six-frame (sync)metamers of every genome, on the GPU, via the public extraction entry point (long-read geometry,
    overlapping 20 kb pieces), a few dozen genomes per call; per-genome de-duplication and the final (value, taxid) order on the
    device when torch is given (thousands of genomes), else numpy."""
    piece, ov = 20000, 32
    p = M.default_params(seq_mode=3, syncmer=params.syncmer, smer_len=params.smer_len)
    vals, tids = [], []
    for g0 in range(0, len(world.genomes), genomes_per_call):
        chunk = world.genomes[g0:g0 + genomes_per_call]
        seqs, owner = [], []
        for gi, (tid, g) in enumerate(chunk):
            for st in range(0, len(g), piece - ov):
                seqs.append(g[st:st + piece])
                owner.append(gi)
                if st + piece >= len(g):
                    break
        offs = np.zeros(len(seqs) + 1, np.uint64)
        offs[1:] = np.cumsum([len(x) for x in seqs])
        bases = np.concatenate(seqs).astype(np.uint8)
        k, _, _ = ctx.extract(p, bases, offs)
        owner = np.asarray(owner, dtype=np.int64)
        seq = ((k["qinfo"] >> np.uint64(32)) & np.uint64(0x1FFFFFFF)).astype(np.int64) - 1       # emission order = piece order = genome order
        first = np.searchsorted(owner[seq], np.arange(len(chunk) + 1))
        if torch is not None:
            kv = torch.from_numpy(np.ascontiguousarray(k["value"]).view(np.int64)).to(dev)
        for gi, (tid, g) in enumerate(chunk):
            if torch is not None:
                v = torch.unique(kv[first[gi]:first[gi + 1]])                                   # signed ascending: [negative values | non-negative values]
                n_neg = int((v < 0).sum().item())
                vals.append((v[n_neg:], v[:n_neg])); tids.append(tid)
            else:
                v = np.unique(k["value"][first[gi]:first[gi + 1]])
                vals.append(v); tids.append(np.full(len(v), tid, np.int32))
    # one strain per species here, so (value, species) pairs are already unique; strain ids rise with the genome order,
    # so a STABLE sort by value of the genome-major concatenation is (value, taxid) order
    if torch is not None:
        # unsigned 64-bit order = the non-negative int64 values ascending, then the negative ones ascending; the halves are
        # sorted separately (torch.sort and boolean masks take < 2^31 elements per call)
        out_v, out_t = [], []
        for half in (0, 1):
            hv = torch.cat([x[half] for x in vals])
            ht = torch.cat([torch.full((len(x[half]),), tid, dtype=torch.int32, device=dev) for x, tid in zip(vals, tids)])
            if len(hv) >= 2**31:
                raise SystemExit(f"{len(hv)} genome-derived metamers in one sign half: more than one torch.sort call takes")
            order = torch.sort(hv, stable=True).indices
            out_v.append(hv[order].cpu().numpy().view(np.uint64)); out_t.append(ht[order].cpu().numpy())
            del hv, ht, order
        del vals
        torch.cuda.empty_cache()
        return np.concatenate(out_v), np.concatenate(out_t)
    vals = np.concatenate(vals); tids = np.concatenate(tids)
    order = np.lexsort((tids, vals))
    return vals[order], tids[order]


def candidate_closure(torch, d_values, d_info, T, q_values):
    """Every target of the flat device index whose amino-acid part equals that of a query metamer, plus the index's true last
    entry -- computed with torch.searchsorted on the flat arrays, i.e. by nothing of the library under test.  Matches(q) depends on
    no other target (C(q) = {t < T-1 : AA(t) = AA(q)}), so the oracle on this sub-database must give every read of
    the sample the answer the GPU gives against the whole index.  Returns (values u64, taxids i32) as numpy arrays."""
    dev = d_values.device
    aa = np.unique(np.asarray(q_values, dtype=np.uint64) >> np.uint64(24))
    v = d_values[:T]
    # the array is sorted as UNSIGNED 64-bit; as int64 it is [non-negative ascending | negative ascending]: find the split by
    # bisection on single elements and search each half with the queries of its sign
    lo, hi = 0, T
    while lo < hi:
        mid = (lo + hi) // 2
        if int(v[mid].item()) >= 0:
            lo = mid + 1
        else:
            hi = mid
    n_pos = lo
    lo_key = (aa << np.uint64(24)).view(np.int64)
    hi_key = (((aa + np.uint64(1)) << np.uint64(24)) - np.uint64(1)).view(np.int64)       # last value of the amino-acid part (no wrap at the top)
    neg = lo_key < 0
    starts = np.empty(len(aa), np.int64); ends = np.empty(len(aa), np.int64)
    for sel, base, seg in ((~neg, 0, v[:n_pos]), (neg, n_pos, v[n_pos:])):
        if not sel.any():
            continue
        a = torch.searchsorted(seg, torch.from_numpy(lo_key[sel]).to(dev), right=False) + base
        b = torch.searchsorted(seg, torch.from_numpy(hi_key[sel]).to(dev), right=True) + base
        starts[sel] = a.cpu().numpy(); ends[sel] = b.cpu().numpy()
    cnt = ends - starts
    keep = cnt > 0
    starts, cnt = starts[keep], cnt[keep]
    tot = int(cnt.sum())
    first = np.zeros(len(cnt) + 1, np.int64); np.cumsum(cnt, out=first[1:])
    idx = np.repeat(starts - first[:-1], cnt) + np.arange(tot, dtype=np.int64)
    if tot == 0 or idx[-1] != T - 1:
        idx = np.append(idx, T - 1)                      # the entry the `t < T-1` rule excludes must be the sub-database's last one too
    di = torch.from_numpy(idx).to(dev)
    cv = d_values[di].cpu().numpy().view(np.uint64); ct = d_info[di].cpu().numpy().view(np.int32) & np.int32(0x7FFFFFFF)
    assert (cv[1:] >= cv[:-1]).all()
    return cv, ct


def gen_reads(torch, dev, world, n_reads, read_len, frac_random, err, seed, paired=False, frag_len=400):
    """Reads sampled from the genomes (both strands, substitutions) + random
    reads, generated on the device so that the inputs are HBM-resident.
    paired: fragments of frag_len bases, mate 1 = its first read_len bases, mate 2 = the first read_len bases of its
    reverse complement (BASELINE.json configs[3] shape); returns (bases1, offs, bases2)."""
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    G = torch.from_numpy(np.concatenate([x for _, x in world.genomes]).astype(np.uint8)).to(dev)
    lens = torch.tensor([len(x) for _, x in world.genomes], device=dev, dtype=torch.int64)
    starts = torch.cumsum(lens, 0) - lens
    comp = torch.full((256,), ord("N"), dtype=torch.uint8, device=dev)
    for a, b in zip(b"ACGT", b"TGCA"):
        comp[a] = b
    acgt = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=dev)
    out = torch.empty(n_reads * read_len, dtype=torch.uint8, device=dev)
    out2 = torch.empty(n_reads * read_len, dtype=torch.uint8, device=dev) if paired else None
    span = frag_len if paired else read_len
    ar = torch.arange(span, device=dev, dtype=torch.int64)
    chunk = 1_000_000
    for c0 in range(0, n_reads, chunk):
        n = min(chunk, n_reads - c0)
        gi = torch.randint(0, len(lens), (n,), generator=g, device=dev)
        st = (torch.rand(n, generator=g, device=dev) * (lens[gi] - span + 1).float()).long().clamp_(min=0) + starts[gi]
        r = G[st[:, None] + ar[None, :]]
        rc = torch.rand(n, generator=g, device=dev) < 0.5
        r = torch.where(rc[:, None], comp[r.long()].flip(1), r)
        sub = torch.rand(n, span, generator=g, device=dev) < err
        r = torch.where(sub, acgt[torch.randint(0, 4, (n, span), generator=g, device=dev)], r)
        rnd = torch.rand(n, generator=g, device=dev) < frac_random
        r = torch.where(rnd[:, None], acgt[torch.randint(0, 4, (n, span), generator=g, device=dev)], r)
        if paired:
            out[c0 * read_len:(c0 + n) * read_len] = r[:, :read_len].reshape(-1)
            out2[c0 * read_len:(c0 + n) * read_len] = comp[r.long()].flip(1)[:, :read_len].reshape(-1)
        else:
            out[c0 * read_len:(c0 + n) * read_len] = r.reshape(-1)
    offs = torch.arange(n_reads + 1, device=dev, dtype=torch.int64) * read_len
    if paired:
        return out, offs, out2
    return out, offs


def compare_with_oracle(M, res, tt, tc, R):
    """GPU per-read results (compacted taxcnt lists) against the oracle's answer R for the same reads: taxon, classified
    flag, score on the fp32 bit pattern, query lengths, and the taxID:match_count lists.  Reads the oracle flags as
    std::sort-ambiguous are excluded and counted.  Returns a dict for the JSON line."""
    ro = R["results"]
    amb = ro["flag"] != 0
    bad = (res["classification"] != ro["classification"]) | (res["is_classified"] != ro["is_classified"]) | \
          (res["score"].view(np.uint32) != ro["score"].view(np.uint32)) | (res["qlen"] != ro["qlen"]) | (res["qlen2"] != ro["qlen2"]) | \
          (res["n_taxcnt"] != ro["n_taxcnt"])
    bad &= ~amb
    # taxID:match_count lists of the reads that agree so far (equal lengths there): gather both sides by their offsets
    ok = np.flatnonzero(~amb & ~bad)
    n = ro["n_taxcnt"][ok].astype(np.int64)
    tot = int(n.sum())
    first = np.zeros(len(ok) + 1, np.int64); np.cumsum(n, out=first[1:])
    within = np.arange(tot, dtype=np.int64) - np.repeat(first[:-1], n)
    ig = np.repeat(res["taxcnt_off"][ok].astype(np.int64), n) + within
    io = np.repeat(ro["taxcnt_off"][ok].astype(np.int64), n) + within
    diff = (tt[ig] != R["tc_tax"][io]) | (tc[ig] != R["tc_cnt"][io])
    list_bad = int(len(np.unique(np.repeat(np.arange(len(ok)), n)[diff])))
    return dict(reads=int(len(ro)), mismatches=int(bad.sum()) + list_bad, ambiguous_excluded=int(amb.sum()),
                classified=int((ro["is_classified"] != 0).sum()),
                checked="classification, is_classified, score (fp32 bits), query lengths, taxID:match_count lists")


def cpu_baseline_and_parity(ctx, M, torch, dev, world, real_v, real_t, params, taxdir, d_bases, d_bases2, read_len, n_sample, n_filler_small, seed,
                            run_cpu=True):
    """(a) cpu_baseline: the oracle (CPU restatement of the reference algorithm) on a bounded sample -- the first
    n_sample reads of this rank against an index built by the same generator with fewer filler metamers -- on all host
    cores and on one.  (b) parity_sample: the SAME reads against the SAME small index through the benchmarked entry
    points (mtb_synth_index -> mtb_index_from_device -> mtb_classify_batch_device, device-resident inputs), compared
    with the oracle's answer read by read.  Outside the timed region."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from helpers import Oracle, default_params as odp
    orc = Oracle()
    T = n_filler_small + len(real_v)
    dv = torch.empty(T, dtype=torch.int64, device=dev); di = torch.empty(T, dtype=torch.int32, device=dev)
    n = ctx.synth_index(seed, n_filler_small, world.filler_tax_lo, world.filler_tax_hi, real_v, real_t, dv.data_ptr(), di.data_ptr())
    vals = dv[:n].cpu().numpy().view(np.uint64); tids = di[:n].cpu().numpy()
    d = tempfile.mkdtemp(prefix="mtb_cpu_")
    op = odp(seq_mode=params.seq_mode, syncmer=params.syncmer, smer_len=params.smer_len)
    orc.write_db(d, vals, tids, op)
    tax = orc.load_taxonomy(taxdir)
    db = orc.open_db(d, tax, op)
    paired = params.seq_mode == 2
    bases = d_bases[: n_sample * read_len].cpu().numpy()
    offs = (np.arange(n_sample + 1, dtype=np.uint64) * np.uint64(read_len))
    bases2 = d_bases2[: n_sample * read_len].cpu().numpy() if paired else None
    offs2 = offs if paired else None
    ncores = os.cpu_count() or 1
    cold_s = None
    if run_cpu and drop_page_cache():       # first run with the database files out of the page cache (they were just written)
        tc0 = time.perf_counter()
        orc.classify_batch(db, tax, op, bases, offs, bases2, offs2, threads=ncores)
        cold_s = time.perf_counter() - tc0
    nw = min(n_sample, 20000)      # untimed: creates the OpenMP thread pool
    orc.classify_batch(db, tax, op, bases[: nw * read_len], offs[: nw + 1], bases2[: nw * read_len] if paired else None, offs[: nw + 1] if paired else None, threads=ncores)
    t0 = time.perf_counter()
    R = orc.classify_batch(db, tax, op, bases, offs, bases2, offs2, threads=ncores)
    dt = time.perf_counter() - t0
    stage_s = dict(orc.last_stage_s); oracle_counts = dict(orc.last_counts)
    cpu = None
    if run_cpu:
        # the single-thread figure on a quarter of the sample (same code, threads=1)
        n1 = max(1, min(n_sample // 4, 100000))
        t1 = time.perf_counter()
        orc.classify_batch(db, tax, op, bases[: n1 * read_len], offs[: n1 + 1], bases2[: n1 * read_len] if paired else None, offs[: n1 + 1] if paired else None, threads=1)
        dt1 = time.perf_counter() - t1
        cls = int((R["results"]["is_classified"] != 0).sum())
        cpu = dict(value=n_sample / dt / 1e6, unit="Mreads/s", cores=ncores, kind="port", cpu_model=cpu_model(),
                   single_thread_value=n1 / dt1 / 1e6, stage_seconds={k: round(v, 3) for k, v in stage_s.items()},
                   cold_cache_first_run_s=cold_s, cold_cache_note=("database files dropped from the page cache before the first run (includes creating the OpenMP pool)"
                                                                   if cold_s is not None else "could not drop the page cache: warm runs only"),
                   numa=numa_layout(), index_targets=int(n), index_file_bytes=int(sum(os.path.getsize(os.path.join(d, f)) for f in ("diffIdx", "info"))),
                   sample=f"{n_sample} x {'2 x ' if paired else ''}{read_len} bp reads vs {n} target metamers ({len(real_v)} genome-derived + {n_filler_small} filler; "
                          f"{n * 12 / 2**20:.0f} MiB flat), oracle/liboracle.so with OpenMP on {ncores} threads, {dt:.1f} s "
                          f"(1 thread on {n1} reads: {dt1:.1f} s), {cls} classified")
    # ---- the same sample through the benchmarked GPU path ----
    taxid_list = np.concatenate([np.unique(real_t), np.arange(world.filler_tax_lo, world.filler_tax_hi + 1, dtype=np.int32)])
    small = ctx.index_from_device(dv.data_ptr(), di.data_ptr(), n, taxdir, taxid_list, params)
    s_res = torch.empty(n_sample * 24, dtype=torch.uint8, device=dev)
    s_cap = n_sample * (20 + read_len // 9) * (2 if paired else 1) + 1024
    s_tt = torch.empty(s_cap, dtype=torch.int32, device=dev); s_tc = torch.empty(s_cap, dtype=torch.int32, device=dev)
    s_offs = torch.arange(n_sample + 1, device=dev, dtype=torch.int64) * read_len
    ntc = ctx.classify_batch_device(small, params, d_bases.data_ptr(), s_offs.data_ptr(), d_bases2.data_ptr() if paired else 0,
                                    s_offs.data_ptr() if paired else 0, n_sample, n_sample * read_len * (2 if paired else 1),
                                    s_res.data_ptr(), s_tt.data_ptr(), s_tc.data_ptr(), s_cap)
    torch.cuda.synchronize()
    res = np.frombuffer(s_res.cpu().numpy().tobytes(), dtype=M.result_dt)
    g_res, g_tt, g_tc = M.compact_taxcnt(res, s_tt[:ntc].cpu().numpy(), s_tc[:ntc].cpu().numpy().view(np.uint32))
    par = compare_with_oracle(M, g_res, g_tt, g_tc, R)
    par["index"] = f"{n} target metamers via mtb_synth_index -> mtb_index_from_device; reads via mtb_classify_batch_device (device-resident)"
    par["matches"] = int(ctx.last_stats().n_matches); par["oracle_matches"] = int(oracle_counts["matches"])
    if par["matches"] != par["oracle_matches"]:
        par["mismatches"] += 1
    small.close()
    del dv, di
    return cpu, par


def parity_full_index(ctx, M, torch, dev, index, params, taxdir, d_bases, d_bases2, read_len, n_s, closure, T):
    """The first n_s reads of the timed batch through the benchmarked entry point against THE TIMED INDEX ITSELF (sealed, packed
    words, depth-7 directory, > 2^32 targets), compared read by read with the oracle run on the candidate closure of those reads
    (candidate_closure above, taken from the flat arrays before they were packed).  Outside the timed region."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from helpers import Oracle, default_params as odp
    orc = Oracle()
    cv, ct = closure
    d = tempfile.mkdtemp(prefix="mtb_closure_")
    op = odp(seq_mode=params.seq_mode, syncmer=params.syncmer, smer_len=params.smer_len)
    orc.write_db(d, cv, ct, op)
    tax = orc.load_taxonomy(taxdir)
    db = orc.open_db(d, tax, op)
    paired = params.seq_mode == 2
    offs_all = d_bases_offsets(torch, dev, n_s, read_len)
    bases = d_bases[: n_s * read_len].cpu().numpy()
    offs = np.arange(n_s + 1, dtype=np.uint64) * np.uint64(read_len)
    bases2 = d_bases2[: n_s * read_len].cpu().numpy() if paired else None
    R = orc.classify_batch(db, tax, op, bases, offs, bases2, offs if paired else None, threads=os.cpu_count() or 1)
    oracle_counts = dict(orc.last_counts)
    s_res = torch.empty(n_s * 24, dtype=torch.uint8, device=dev)
    s_cap = n_s * (20 + read_len // 9) * (2 if paired else 1) + 1024
    s_tt = torch.empty(s_cap, dtype=torch.int32, device=dev); s_tc = torch.empty(s_cap, dtype=torch.int32, device=dev)
    ntc = ctx.classify_batch_device(index, params, d_bases.data_ptr(), offs_all.data_ptr(), d_bases2.data_ptr() if paired else 0,
                                    offs_all.data_ptr() if paired else 0, n_s, n_s * read_len * (2 if paired else 1),
                                    s_res.data_ptr(), s_tt.data_ptr(), s_tc.data_ptr(), s_cap)
    torch.cuda.synchronize()
    res = np.frombuffer(s_res.cpu().numpy().tobytes(), dtype=M.result_dt)
    g_res, g_tt, g_tc = M.compact_taxcnt(res, s_tt[:ntc].cpu().numpy(), s_tc[:ntc].cpu().numpy().view(np.uint32))
    par = compare_with_oracle(M, g_res, g_tt, g_tc, R)
    par["matches"] = int(ctx.last_stats().n_matches); par["oracle_matches"] = int(oracle_counts["matches"])
    if par["matches"] != par["oracle_matches"]:
        par["mismatches"] += 1
    stt = index.state()
    par["index"] = (f"the timed index itself: {T} targets, directory depth {stt['dir_depth']}, {'packed 8-byte words' if stt['packed'] else 'flat {value, info}'}"
                    f"{', sealed' if stt['sealed'] else ''}; oracle on the candidate closure of the sample's metamers ({len(cv)} targets incl. the index's last entry, "
                    "gathered with torch.searchsorted from the flat arrays before packing)")
    par["closure_targets"] = int(len(cv))
    return par


def d_bases_offsets(torch, dev, n, read_len):
    return torch.arange(n + 1, device=dev, dtype=torch.int64) * read_len


def numa_layout():
    """NUMA nodes of the host and their CPU lists (printed next to the CPU baseline)"""
    out = {}
    try:
        base = "/sys/devices/system/node"
        for d in sorted(os.listdir(base)):
            if d.startswith("node") and d[4:].isdigit():
                out[d] = open(os.path.join(base, d, "cpulist")).read().strip()
    except OSError:
        pass
    return out


def drop_page_cache():
    """cold-cache run of the CPU baseline: needs root and a writable /proc/sys/vm/drop_caches"""
    try:
        os.sync()
        with open("/proc/sys/vm/drop_caches", "w") as f:
            f.write("3\n")
        return True
    except OSError:
        return False


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def silence_other_ranks(rank):
    """torch.distributed.run interleaves every rank's stdout: only rank 0 may write there (C libraries included)"""
    if rank != 0:
        os.dup2(os.open(os.devnull, os.O_WRONLY), 1)


def finish(dist, line):
    """Rank 0's JSON line is the LAST thing on stdout: collectives are torn down first and whatever C libraries (RCCL prints its
    library path through C stdio, which is block-buffered on a pipe) left in the C-level buffer is flushed before it."""
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    sys.stdout.flush()
    try:
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except OSError:
        pass
    if line is not None:
        print(line, flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--reads", type=int, default=10_000_000, help="reads per GPU per step")
    ap.add_argument("--read-len", type=int, default=150)
    ap.add_argument("--targets", type=float, default=16e9,
                    help="filler metamers in the synthetic index (per GPU, replicated); 16 G = SURVEY 8(d)'s GTDB-scale planning size, 192 GB flat")
    ap.add_argument("--species", type=int, default=24,
                    help="genomes the reads are drawn from (and whose metamers are in the index); >= 200 takes the device-side generator: "
                         "the genome-diversity run is --species 2400 --fixed-total (0.6 x coverage instead of 62 x)")
    ap.add_argument("--fixed-total", action="store_true", help="--targets is the TOTAL number of target metamers (filler = total - genome-derived)")
    ap.add_argument("--full-parity-reads", type=int, default=32768, help="reads of the parity check against the timed index itself (0 = off)")
    ap.add_argument("--genome-len", type=int, default=1_000_000)
    ap.add_argument("--filler-species", type=int, default=130_000)
    ap.add_argument("--cpu-reads", type=int, default=2_000_000, help="reads of the CPU-baseline / parity sample (the first reads of rank 0's batch)")
    ap.add_argument("--cpu-targets", type=float, default=1e9,
                    help="filler metamers of the CPU baseline's / parity sample's index (1 G: 9 GB of diffIdx + info on the host, the oracle streams it "
                         "from every split checkpoint; the whole bench run then takes about two minutes, 16e6 brings it back to 35 s)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the timed CPU baseline (the parity sample still runs the oracle)")
    ap.add_argument("--no-parity", action="store_true", help="skip the oracle comparison of the benchmarked path (and the CPU baseline)")
    ap.add_argument("--streams", type=int, default=1, help="HIP streams a batch is pipelined over inside the library")
    ap.add_argument("--seq-mode", type=int, default=1, choices=[1, 2, 3],
                    help="1 = short single-end (configs[1]); 2 = paired-end, --reads pairs of 2 x --read-len (configs[3] shape); 3 = long reads (configs[2])")
    ap.add_argument("--partitioned", action="store_true",
                    help="SURVEY 8(e) row 2: every rank owns one value range of the index; metamers and matches travel by all-to-all "
                         "(functional/perf check of that path; the default is the replicated index)")
    ap.add_argument("--prealloc", action="store_true", help="experiment: grow the context's workspace on a tiny index BEFORE the big index is allocated")
    ap.add_argument("--no-seal", action="store_true", help="keep the flat {value, info} arrays next to the packed state (mtb_index_seal not called)")
    ap.add_argument("--seed", type=int, default=1234)
    ap.add_argument("--dist-backend", default="nccl", help="testing only: gloo lets several ranks share one GPU (RCCL refuses duplicate devices)")
    ap.add_argument("--shared-gpu", action="store_true", help="testing only: every rank uses cuda:0")
    args = ap.parse_args()

    import torch  # before libmtb: both must share one HIP runtime (libamdhip64.so.7)
    rank = int(os.environ.get("RANK", "0")); world_size = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    silence_other_ranks(rank)
    if world_size != args.gpus:
        log(f"warning: WORLD_SIZE={world_size} but --gpus {args.gpus}")
    if args.shared_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world_size > 1:
        import torch.distributed as dist
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world_size, device_id=dev)
        else:
            dist.init_process_group(args.dist_backend, rank=rank, world_size=world_size)
    elif args.partitioned:
        import torch.distributed as dist
        dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{29400 + os.getpid() % 500}", rank=0, world_size=1, device_id=dev)
    import metabuli_amd as M
    ctx = M.Context(local_rank)
    ctx.set_streams(args.streams)
    ctx.set_placement_probe(True)      # this process has allocated and freed > 200 GB through torch by now: see mtb_ctx_set_placement_probe (include/mtb.h)
    params = M.default_params(seq_mode=args.seq_mode, syncmer=1, smer_len=5)

    t_setup = time.perf_counter()
    big_world = args.species >= 200
    world = build_world_fast(torch, dev, args.seed, args.species, args.genome_len, args.filler_species) if big_world else \
        build_world(args.seed, args.species, args.genome_len, args.filler_species)
    taxdir = tempfile.mkdtemp(prefix="mtb_tax_")
    world.tax.write(taxdir)
    real_v, real_t = extract_targets(ctx, M, world, params, torch if big_world else None, dev)
    n_filler = int(args.targets) - (len(real_v) if args.fixed_total else 0)
    log(f"[rank {rank}] world: {len(world.genomes)} genomes x {args.genome_len} bp, {len(real_v)} genome-derived target metamers ({time.perf_counter()-t_setup:.1f}s)")
    T_cap = n_filler + len(real_v)
    if args.prealloc:
        nf0 = 1_000_000
        v0 = torch.empty(nf0 + len(real_v), dtype=torch.int64, device=dev); i0 = torch.empty(nf0 + len(real_v), dtype=torch.int32, device=dev)
        T0 = ctx.synth_index(args.seed, nf0, world.filler_tax_lo, world.filler_tax_hi, real_v, real_t, v0.data_ptr(), i0.data_ptr())
        tl0 = np.concatenate([np.unique(real_t), np.arange(world.filler_tax_lo, world.filler_tax_hi + 1, dtype=np.int32)])
        ix0 = ctx.index_from_device(v0.data_ptr(), i0.data_ptr(), T0, taxdir, tl0, params)
        b0, o0 = gen_reads(torch, dev, world, args.reads, args.read_len, 0.10, 0.005, args.seed + 17 * (rank + 1))
        r0 = torch.empty(args.reads * 24, dtype=torch.uint8, device=dev)
        cap0 = args.reads * (20 + args.read_len // 9) + 1024
        t0_ = torch.empty(cap0, dtype=torch.int32, device=dev); c0_ = torch.empty(cap0, dtype=torch.int32, device=dev)
        for _ in range(2):
            ctx.classify_batch_device(ix0, params, b0.data_ptr(), o0.data_ptr(), 0, 0, args.reads, args.reads * args.read_len, r0.data_ptr(), t0_.data_ptr(), c0_.data_ptr(), cap0)
        torch.cuda.synchronize()
        del ix0, v0, i0, b0, o0, r0, t0_, c0_
        torch.cuda.empty_cache()
    free, total = torch.cuda.mem_get_info(dev)
    need = T_cap * 12 + args.reads * (args.read_len + 8)
    if need > free * 0.9:
        raise SystemExit(f"index of {T_cap} targets needs {need/2**30:.0f} GiB, only {free/2**30:.0f} GiB free")
    d_values = torch.empty(T_cap, dtype=torch.int64, device=dev)
    d_info = torch.empty(T_cap, dtype=torch.int32, device=dev)
    T = ctx.synth_index(args.seed, n_filler, world.filler_tax_lo, world.filler_tax_hi, real_v, real_t, d_values.data_ptr(), d_info.data_ptr())
    taxid_list = np.concatenate([np.unique(real_t), np.arange(world.filler_tax_lo, world.filler_tax_hi + 1, dtype=np.int32)])
    index = ctx.index_from_device(d_values.data_ptr(), d_info.data_ptr(), T, taxdir, taxid_list, params)
    d_bases2 = None
    if args.seq_mode == 2:
        d_bases, d_offs, d_bases2 = gen_reads(torch, dev, world, args.reads, args.read_len, 0.10, 0.005, args.seed + 17 * (rank + 1), paired=True)
    else:
        d_bases, d_offs = gen_reads(torch, dev, world, args.reads, args.read_len, 0.10, 0.005, args.seed + 17 * (rank + 1))
    # candidate closure of the parity sample, from the flat arrays and before anything packs them (parity_full_index below)
    closure, n_full = None, 0
    if rank == 0 and world_size == 1 and not args.no_parity and not args.partitioned and args.full_parity_reads > 0:
        n_full = min(args.reads, args.full_parity_reads if args.seq_mode != 3 else max(1, args.full_parity_reads * 150 // args.read_len))
        sb = d_bases[: n_full * args.read_len].cpu().numpy()
        so = np.arange(n_full + 1, dtype=np.uint64) * np.uint64(args.read_len)
        sk, _, _ = ctx.extract(params, sb, so, d_bases2[: n_full * args.read_len].cpu().numpy() if d_bases2 is not None else None, so if d_bases2 is not None else None)
        t_c = time.perf_counter()
        closure = candidate_closure(torch, d_values, d_info, T, sk["value"])
        log(f"[rank 0] candidate closure of {n_full} reads ({len(sk)} metamers): {len(closure[0])} targets ({time.perf_counter()-t_c:.1f}s)")
        del sk
    sealed = False
    if not args.partitioned and not args.no_seal:
        # dedicate the index to the fused path: packed 8-byte target words under the amino-acid directory; the info array lent to
        # the library is no longer needed by it and is freed here (64 GB at 16 G targets)
        try:
            index.seal(); sealed = True
            del d_info
            torch.cuda.empty_cache()
        except M.MtbError as e:
            log(f"[rank {rank}] index not sealed: {e}")
    n_bases_step = args.reads * args.read_len * (2 if args.seq_mode == 2 else 1)
    d_res = torch.empty(args.reads * 24, dtype=torch.uint8, device=dev)
    tc_cap = args.reads * (20 + args.read_len // 9) * (2 if args.seq_mode == 2 else 1) + 1024
    d_tt = torch.empty(tc_cap, dtype=torch.int32, device=dev); d_tc = torch.empty(tc_cap, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    log(f"[rank {rank}] setup {time.perf_counter()-t_setup:.1f}s: T={T} ({len(real_v)} genome-derived), reads={args.reads}x{args.read_len}")

    part = None
    if args.partitioned:
        # range r = [bounds[r], bounds[r+1]) cut at amino-acid-part boundaries of the resident array; every rank
        # keeps a device view of its own range only (the full array stays allocated: this mode measures the
        # exchange path, not the capacity gain)
        from metabuli_amd import parallel
        AAM = ~0xFFFFFF
        bounds = np.zeros(world_size, np.uint64)
        for r in range(1, world_size):
            bounds[r] = np.uint64(int(d_values[(T * r) // world_size].item()) & AAM & (2**64 - 1))
        hi = int(bounds[rank + 1]) if rank + 1 < world_size else 2**64 - 1
        part = index.slice(int(bounds[rank]), hi, rank == world_size - 1)
        stages = parallel.GpuStages(ctx, part, params, dev)
        stages.set_reads(d_bases, d_offs, args.reads)
        log(f"[rank {rank}] partitioned: range {rank} holds {part.num_targets} of {T} targets")
        last = {}

    def step():
        if part is not None:
            last["res"] = parallel.classify_partitioned(stages, bounds, dist)
            return 0
        return ctx.classify_batch_device(index, params, d_bases.data_ptr(), d_offs.data_ptr(),
                                         d_bases2.data_ptr() if d_bases2 is not None else 0, d_offs.data_ptr() if d_bases2 is not None else 0,
                                         args.reads, n_bases_step, d_res.data_ptr(), d_tt.data_ptr(), d_tc.data_ptr(), tc_cap)

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    sub_batches_timed = ctx.last_sub_batches if part is None else 0      # of the last timed step (the profiled step and the parity sample overwrite the counter)
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    st = ctx.last_stats()

    if part is not None:
        res = last["res"][0]
        frac_cls = float((res["is_classified"] != 0).mean())
        log(f"[rank {rank}] partitioned step: classified {frac_cls:.4f}")
        if rank == 0:
            value = args.reads * world_size * args.steps / dt / 1e6
            line = json.dumps(dict(metric="Mreads/s classified (metabuli classify hot path, reads + index resident in HBM)",
                                  value=value, unit="Mreads/s", n_gpus=world_size, steps=args.steps, warmup=args.warmup,
                                  ms_per_step=dt / args.steps * 1e3, higher_is_better=True, scaling="weak", vs_baseline=None,
                                  dtype="u64", data="synthetic",
                                  config=dict(workload=f"{args.reads/1e6:g}M x {args.read_len} bp reads per GPU vs {T/1e9:.2f} G metamers "
                                                       f"range-partitioned over {world_size} GPU(s) (SURVEY 8(e) row 2)",
                                              reads_per_gpu=args.reads, read_len=args.read_len, targets=int(T), seq_mode=args.seq_mode,
                                              classified_fraction=frac_cls,
                                              parallelism=f"index range-partitioned x{world_size}, 2 all-to-all per batch; results to host"),
                                  roofline=None, cpu_baseline=None))
        finish(dist, line if rank == 0 else None)
        return

    # one extra, untimed, profiled step on ONE stream (kernels not overlapped, full-batch launches):
    # HIP events on the library's stream around every kernel launch
    ctx.set_streams(1)
    ctx.set_profiling(True)
    step()
    ps = ctx.last_stats()
    ctx.set_profiling(False)
    ctx.set_streams(args.streams)
    kern = {M.KERNEL_NAMES[i]: dict(ms=float(ps.ms_kernel[i]), launches=int(ps.n_launch[i])) for i in range(len(M.KERNEL_NAMES))}
    # index-side working set of that step's directory join (diagnostic kernel over the step's sorted metamers)
    footprint = None
    try:
        fp = ctx.join_footprint(index)
        least = fp.target_sectors * 64 + fp.dir_sectors * 64 + 16 * fp.n_queries
        footprint = dict(query_metamers=int(fp.n_queries), distinct_buckets=int(fp.distinct_buckets), buckets=int(fp.n_buckets),
                         directory_sectors_64B=int(fp.dir_sectors), target_sectors_64B=int(fp.target_sectors),
                         target_sectors_total=int(fp.n_targets * 8 // 64), target_fraction_touched=fp.target_sectors * 64 / max(1, fp.n_targets * 8),
                         least_fetch_bytes=int(least),
                         note="distinct 64-byte sectors the batch's queries address (bucket spans of the target array, directory words) + 16 B per query: "
                              "the least k_join_dir can fetch for this batch; 12 x T of the contract formula is a streaming-merge figure this kernel never pays")
    except M.MtbError as e:
        log(f"[rank {rank}] no join footprint: {e}")
    Kq, Mm, N, L = ps.n_kmers, ps.n_matches, ps.n_reads, ps.n_bases
    # algorithmic bytes of one whole step per kernel (SURVEY.md 8(d) per-stage split);
    # a step launches every kernel once per stream (and per radix pass): bytes per launch = total / launches
    alg_step = {"extract_count": L, "extract_emit": L + 16 * Kq, "radix_hist": 16 * Kq, "radix_scatter": 32 * Kq,
                "join": 16 * Kq + 24 * Mm, "regroup": 48 * Mm, "score": 24 * Mm + 16 * N, "score_fast": 24 * Mm + 16 * N}
    if params.kmer_format == 2 and kern["radix_hist"]["launches"]:
        # the fused path's histograms read the 2-byte digit side arrays and write the 4-byte tile table, not the 16-byte records
        alg_step["radix_hist"] = kern["radix_hist"]["launches"] * (2 * Kq + 4 * 512 * ((Kq + 4095) // 4096))
    alg = {k: v / max(1, kern[k]["launches"]) for k, v in alg_step.items()}
    if kern.get("score_fast", {}).get("launches"):         # the two scoring kernels share the reads
        gfrac = ps.n_generic_reads / max(1, N)
        alg["score"] *= gfrac; alg["score_fast"] *= 1.0 - gfrac
    alg["join"] += 12 * ps.n_targets          # the 12*T_span term is paid by every launch (one per HBM-budgeted sub-batch): each spans the whole index
    dom = max((k for k in alg), key=lambda k: kern[k]["ms"])
    avg_ms = kern[dom]["ms"] / max(1, kern[dom]["launches"])
    achieved = alg[dom] / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
    # HBM traffic of the dominant kernel from the committed PMC passes (rocprofv3 cannot run inside this process);
    # only when they were taken on this very workload
    traffic, traffic_note = None, "no PMC passes for this workload (profiles/pmc_traffic.json)"
    try:
        pj = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
        wl = pj["workload"]
        if (wl["reads"], wl["read_len"], wl["targets"], wl["seq_mode"]) == (args.reads, args.read_len, int(ps.n_targets), args.seq_mode) and dom in pj["kernels"]:
            kk = pj["kernels"][dom]
            traffic = (2.0 * kk.get("fetch_size_kb", 0.0) + kk.get("write_size_kb", 0.0)) * 1024.0
            traffic_note = f"{pj['source']}: {pj['correction']}"
    except (OSError, KeyError, ValueError):
        pass
    # every kernel of the step against the same peak (informational; `roofline` below is the dominant one)
    roofline_all = {k: dict(ms=round(kern[k]["ms"], 3), launches=kern[k]["launches"], algorithmic_gb_per_launch=round(alg[k] / 1e9, 3),
                            achieved_gb_s=round(alg[k] / (kern[k]["ms"] / max(1, kern[k]["launches"]) * 1e-3) / 1e9, 1),
                            frac=round(alg[k] / (kern[k]["ms"] / max(1, kern[k]["launches"]) * 1e-3) / 1e9 / 8000.0, 4))
                    for k in alg if kern[k]["ms"] > 0}
    effective = (traffic / (avg_ms * 1e-3) / 1e9 / 8000.0) if (traffic and avg_ms > 0) else None
    roofline = dict(bound="hbm", kernel=dom, achieved=achieved, peak=8000.0, unit="GB/s", frac=achieved / 8000.0, traffic=traffic, traffic_note=traffic_note,
                    effective=effective,
                    effective_note="traffic / avg_launch_ms / peak: the fraction of HBM bandwidth the kernel really moves (PMC bytes, not the contract's algorithmic bytes)",
                    avg_launch_ms=avg_ms, launches=kern[dom]["launches"], algorithmic_bytes_per_launch=alg[dom],
                    footprint=footprint if dom == "join" else None,
                    note="per-kernel durations from HIP events around every launch of one extra step; traffic = HBM bytes per launch from the PMC passes; "
                         "`frac` follows SURVEY 8(d)'s formula (16 Kq + 12 T + 24 M for the join) and is NOT a bandwidth fraction for the directory join: see `effective` and `footprint`")

    # sanity of the timed output: fraction of reads classified
    res = np.frombuffer(d_res.cpu().numpy().tobytes(), dtype=M.result_dt)
    frac_cls = float((res["is_classified"] != 0).mean())
    log(f"[rank {rank}] stage ms: extract {st.ms_extract:.1f} sort {st.ms_sort:.1f} join {st.ms_join:.1f} regroup {st.ms_regroup:.1f} "
        f"segsort {st.ms_segsort:.2f} score {st.ms_score:.1f} total {st.ms_total:.1f}; classified {frac_cls:.4f}")
    if frac_cls < 0.5:   # 90 % of the reads come from genomes that are in the index
        raise SystemExit(f"sanity check failed: only {frac_cls:.4f} of the reads were classified")

    cpu, parity = None, None
    if rank == 0 and world_size == 1 and not args.no_parity and not big_world:       # (the small-index sample would need all 2.4 G genome-derived entries: the diversity run keeps the check against the timed index only)
        cpu, parity = cpu_baseline_and_parity(ctx, M, torch, dev, world, real_v, real_t, params, taxdir, d_bases, d_bases2, args.read_len,
                                              min(args.cpu_reads, args.reads), int(args.cpu_targets), args.seed, run_cpu=not args.no_cpu)
        log(f"[rank 0] parity sample: {parity}")
        if parity["mismatches"]:
            raise SystemExit(f"parity check failed: {parity}")
    parity_full = None
    if closure is not None:
        parity_full = parity_full_index(ctx, M, torch, dev, index, params, taxdir, d_bases, d_bases2, args.read_len, n_full, closure, T)
        log(f"[rank 0] parity against the timed index: {parity_full}")
        if parity_full["mismatches"]:
            raise SystemExit(f"parity check against the timed index failed: {parity_full}")

    if rank == 0:
        total_reads = args.reads * world_size * args.steps
        value = total_reads / dt / 1e6
        out = dict(metric="Mreads/s classified (metabuli classify hot path, reads + index resident in HBM)",
                   value=value, unit="Mreads/s", n_gpus=world_size, steps=args.steps, warmup=args.warmup,
                   ms_per_step=dt / args.steps * 1e3, higher_is_better=True, scaling="weak", vs_baseline=None,
                   dtype="u64", data="synthetic",
                   config=dict(workload=f"{args.reads/1e6:g}M x {'2 x ' if args.seq_mode == 2 else ''}{args.read_len} bp synthetic "
                                        f"{ {1: 'single-end', 2: 'paired-end', 3: 'long'}[args.seq_mode] } reads per GPU vs synthetic "
                                        f"GTDB-scale index of {T/1e9:.2f} G metamers ({T*12/2**30:.0f} GiB flat, replicated per GPU), "
                                        f"syncmer s=5, kmer_format 2 ({ {1: 'BASELINE.json configs[1]', 2: 'BASELINE.json configs[3] shape: paired-end, index replicated, reads sharded', 3: 'BASELINE.json configs[2] shape: long reads'}[args.seq_mode] })",
                               reads_per_gpu=args.reads, read_len=args.read_len, targets=int(T), seq_mode=args.seq_mode,
                               gbp_per_s=value * args.read_len * (2 if args.seq_mode == 2 else 1) / 1e3, query_metamers=int(st.n_kmers), matches=int(st.n_matches),
                               classified_fraction=frac_cls, parallelism=f"reads sharded x{world_size}, index replicated", streams_per_gpu=args.streams,
                               sub_batches_per_step=sub_batches_timed, index_sealed=sealed,
                               index_bytes=int(T * (8 if sealed else 12) + 4 * (21 ** index.state()["dir_depth"] + 1)), species=args.species, genome_len=args.genome_len, reads_scored_by_generic_kernel=int(ps.n_generic_reads), reads_on_ordinal_slots=int(ps.n_slot_reads)),
                   stage_ms=dict(extract=st.ms_extract, sort=st.ms_sort, join=st.ms_join, regroup=st.ms_regroup,
                                 segsort=st.ms_segsort, score=st.ms_score, total=st.ms_total),
                   kernel_ms=kern, roofline=roofline, roofline_all=roofline_all, join_footprint=footprint, cpu_baseline=cpu, parity_sample=parity,
                   parity_full_index=parity_full)
        finish(dist, json.dumps(out))
    else:
        finish(dist, None)


if __name__ == "__main__":
    main()
