"""ctypes wrappers for the CPU oracle (oracle/liboracle.so) and the test-only
host build of the kernel arithmetic (tests/emu/libmtb_emu.so), plus shared
dtypes.  Test infrastructure only."""
from __future__ import annotations

import ctypes as C
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

kmer_dt = np.dtype([("value", "<u8"), ("qinfo", "<u8")])
match_dt = np.dtype([("qinfo", "<u8"), ("target_id", "<i4"), ("species_id", "<i4"), ("dna", "<u4"),
                     ("reh", "<u2"), ("ham", "u1"), ("pad", "u1")])
result_dt = np.dtype([("classification", "<i4"), ("score", "<f4"), ("qlen", "<i4"), ("qlen2", "<i4"),
                      ("is_classified", "u1"), ("flag", "u1"), ("n_taxcnt", "<u2"), ("taxcnt_off", "<u4")])
assert kmer_dt.itemsize == 16 and match_dt.itemsize == 24 and result_dt.itemsize == 24


class Params(C.Structure):
    _fields_ = [("seq_mode", C.c_int32), ("syncmer", C.c_int32), ("smer_len", C.c_int32),
                ("kmer_format", C.c_int32), ("min_cons_cnt", C.c_int32), ("min_cons_cnt_euk", C.c_int32),
                ("min_score", C.c_float), ("min_sp_score", C.c_float), ("tie_ratio", C.c_float),
                ("accession_level", C.c_int32), ("skip_redundancy", C.c_int32)]


def default_params(**kw):
    p = Params(seq_mode=1, syncmer=1, smer_len=5, kmer_format=2, min_cons_cnt=4, min_cons_cnt_euk=9,
               min_score=0.0, min_sp_score=0.0, tie_ratio=0.95, accession_level=0, skip_redundancy=1)
    for k, v in kw.items():
        setattr(p, k, v)
    return p


def qinfo_fields(q):
    q = np.asarray(q, dtype=np.uint64)
    return (q & np.uint64(0xFFFFFFFF)).astype(np.uint32), ((q >> np.uint64(32)) & np.uint64(0x1FFFFFFF)).astype(np.uint32), (q >> np.uint64(61)).astype(np.uint32)


def _ptr(a, t=C.c_void_p):
    if a is None:
        return None
    return a.ctypes.data_as(t)


def build_oracle():
    d = os.path.join(ROOT, "oracle")
    so = os.path.join(d, "liboracle.so")
    src = [os.path.join(d, "oracle.cpp"), os.path.join(d, "oracle.h")]
    if (not os.path.exists(so)) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in src):
        subprocess.check_call(["make", "-C", d, "liboracle.so"], stdout=subprocess.DEVNULL)
    return so


def build_emu():
    d = os.path.join(ROOT, "tests", "emu")
    so = os.path.join(d, "libmtb_emu.so")
    src = [os.path.join(d, "emu.cpp"), os.path.join(ROOT, "metabuli_amd", "csrc", "mtb_core.h"),
           os.path.join(ROOT, "metabuli_amd", "csrc", "mtb_score_par.h"), os.path.join(ROOT, "metabuli_amd", "csrc", "host_db.h"),
           os.path.join(ROOT, "include", "mtb.h")]
    if (not os.path.exists(so)) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in src):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-ffp-contract=off", "-shared", "-o", so,
                               os.path.join(d, "emu.cpp")])
    return so


class Oracle:
    def __init__(self):
        self.lib = C.CDLL(build_oracle())
        L = self.lib
        L.orc_extract_batch.restype = C.c_size_t
        L.orc_match_kmers.restype = C.c_size_t
        L.orc_score.restype = C.c_size_t
        L.orc_diffidx_encode.restype = C.c_size_t
        L.orc_diffidx_decode.restype = C.c_size_t
        L.orc_taxonomy_load.restype = C.c_void_p
        L.orc_db_open.restype = C.c_void_p
        L.orc_db_num_kmers.restype = C.c_size_t
        L.orc_hamming_sum.restype = C.c_uint8
        L.orc_hammings.restype = C.c_uint16
        L.orc_hammings_reverse.restype = C.c_uint16
        for f in ("orc_hamming_sum", "orc_hammings", "orc_hammings_reverse"):
            getattr(L, f).argtypes = [C.c_uint64, C.c_uint64]

    # tables
    def codon_tables(self):
        aa = np.zeros(512, np.int32); num = np.zeros(512, np.int32)
        self.lib.orc_codon_tables(_ptr(aa), _ptr(num))
        return aa.reshape(8, 8, 8), num.reshape(8, 8, 8)

    def base_codes(self):
        f = np.zeros(256, np.uint8); r = np.zeros(256, np.uint8)
        self.lib.orc_base_codes(_ptr(f), _ptr(r))
        return f, r

    def hamming_tables(self):
        lk = np.zeros(64, np.uint8); lut = np.zeros(8 * 64, np.uint16)
        self.lib.orc_hamming_tables(_ptr(lk), _ptr(lut))
        return lk.reshape(8, 8), lut.reshape(8, 64)

    def extract_batch(self, p, bases, offs, bases2=None, offs2=None):
        n = len(offs) - 1
        total = int(offs[-1]) + (int(offs2[-1]) if offs2 is not None else 0)
        cap = max(16, total * 2 + 64)
        out = np.zeros(cap, kmer_dt)
        ql = np.zeros(n, np.int32); ql2 = np.zeros(n, np.int32)
        cnt = self.lib.orc_extract_batch(_ptr(bases), _ptr(offs), _ptr(bases2), _ptr(offs2), C.c_size_t(n),
                                         C.byref(p), _ptr(out), C.c_size_t(cap), _ptr(ql), _ptr(ql2))
        assert cnt <= cap
        return out[:cnt].copy(), ql, ql2

    def sort_kmers(self, k):
        k = k.copy()
        self.lib.orc_sort_kmers(_ptr(k), C.c_size_t(len(k)))
        return k

    def diffidx_encode(self, values):
        values = np.ascontiguousarray(values, dtype=np.uint64)
        out = np.zeros(5 * len(values) + 8, np.uint16)
        n = self.lib.orc_diffidx_encode(_ptr(values), C.c_size_t(len(values)), _ptr(out))
        return out[:n].copy()

    def diffidx_decode(self, d16):
        d16 = np.ascontiguousarray(d16, dtype=np.uint16)
        out = np.zeros(len(d16) + 1, np.uint64)
        n = self.lib.orc_diffidx_decode(_ptr(d16), C.c_size_t(len(d16)), _ptr(out))
        return out[:n].copy()

    def write_db(self, d, values, taxids, p, split_num=4096):
        os.makedirs(d, exist_ok=True)
        values = np.ascontiguousarray(values, dtype=np.uint64)
        taxids = np.ascontiguousarray(taxids, dtype=np.int32)
        rc = self.lib.orc_write_db(d.encode(), _ptr(values), _ptr(taxids), C.c_size_t(len(values)),
                                   C.c_int(split_num), C.byref(p))
        assert rc == 0

    def load_taxonomy(self, d):
        t = self.lib.orc_taxonomy_load(os.path.join(d, "names.dmp").encode(), os.path.join(d, "nodes.dmp").encode(),
                                       os.path.join(d, "merged.dmp").encode())
        assert t
        return C.c_void_p(t)

    def open_db(self, d, tax, p):
        db = self.lib.orc_db_open(d.encode(), tax, C.byref(p))
        assert db
        return C.c_void_p(db)

    def match(self, db, sorted_kmers):
        cap = max(1024, 8 * len(sorted_kmers))
        while True:
            out = np.zeros(cap, match_dt)
            n = self.lib.orc_match_kmers(db, _ptr(sorted_kmers), C.c_size_t(len(sorted_kmers)), _ptr(out), C.c_size_t(cap))
            if n <= cap:
                return out[:n].copy()
            cap = n

    def sort_matches(self, m):
        m = m.copy()
        self.lib.orc_sort_matches(_ptr(m), C.c_size_t(len(m)))
        return m

    def score(self, db, tax, p, sorted_matches, n_reads, ql, ql2):
        res = np.zeros(n_reads, result_dt)
        cap = max(1024, len(sorted_matches) + 16)
        tt = np.zeros(cap, np.int32); tc = np.zeros(cap, np.uint32)
        n = self.lib.orc_score(db, tax, C.byref(p), _ptr(sorted_matches), C.c_size_t(len(sorted_matches)),
                               C.c_size_t(n_reads), _ptr(ql), _ptr(ql2), _ptr(res), _ptr(tt), _ptr(tc), C.c_size_t(cap))
        assert n <= cap
        return res, tt[:n].copy(), tc[:n].copy()

    def classify_batch(self, db, tax, p, bases, offs, bases2=None, offs2=None, threads=1):
        """the whole batch inside the library (no Python copies between the stages; bench.py's cpu_baseline).  Returns the
        per-read results + taxcnt lists, per-stage seconds in last_stage_s, counts in last_counts."""
        n = len(offs) - 1
        res = np.zeros(n, result_dt)
        cap = max(1024, 64 * n)
        self.lib.orc_classify_batch.restype = C.c_size_t
        self.set_threads(threads)
        try:
            while True:
                tt = np.zeros(cap, np.int32); tc = np.zeros(cap, np.uint32)
                st = (C.c_double * 5)(); nk = C.c_size_t(); nm = C.c_size_t()
                w = self.lib.orc_classify_batch(db, tax, C.byref(p), _ptr(bases), _ptr(offs), _ptr(bases2), _ptr(offs2), C.c_size_t(n),
                                                _ptr(res), _ptr(tt), _ptr(tc), C.c_size_t(cap), st, C.byref(nk), C.byref(nm))
                if w <= cap:
                    break
                cap = w
        finally:
            self.set_threads(1)
        self.last_stage_s = dict(zip(("extract", "sort_kmers", "match", "sort_matches", "score"), list(st)))
        self.last_counts = dict(kmers=nk.value, matches=nm.value)
        return dict(results=res, tc_tax=tt[:w].copy(), tc_cnt=tc[:w].copy())

    def set_threads(self, n):
        """host threads of the oracle's OpenMP layer (1 = the serial restatement)"""
        self.lib.orc_set_threads(C.c_int(int(n)))

    def classify(self, db, tax, p, bases, offs, bases2=None, offs2=None, threads=1):
        import time
        self.set_threads(threads)
        try:
            t = [time.perf_counter()]
            k, ql, ql2 = self.extract_batch(p, bases, offs, bases2, offs2); t.append(time.perf_counter())
            ks = self.sort_kmers(k); t.append(time.perf_counter())
            m = self.match(db, ks); t.append(time.perf_counter())
            m = self.sort_matches(m); t.append(time.perf_counter())
            res, tt, tc = self.score(db, tax, p, m, len(offs) - 1, ql, ql2); t.append(time.perf_counter())
        finally:
            self.set_threads(1)
        self.last_stage_s = dict(zip(("extract", "sort_kmers", "match", "sort_matches", "score"), np.diff(t).tolist()))
        return dict(kmers=ks, matches=m, results=res, tc_tax=tt, tc_cnt=tc, qlen=ql, qlen2=ql2)


class Emu:
    """Host build of mtb_core.h (tests only)."""

    def __init__(self):
        self.lib = C.CDLL(build_emu())
        self.lib.emu_extract_batch.restype = C.c_size_t
        self.lib.emu_join.restype = C.c_size_t
        self.lib.emu_score.restype = C.c_size_t
        self.lib.emu_score_par.restype = C.c_size_t

    def tables(self):
        buf = np.zeros(256 + 64 + 32, np.uint8)
        self.lib.emu_tables(_ptr(buf))
        return buf[:256].copy(), buf[256:320].copy(), buf[320:352].view(np.uint32).copy()

    def extract_batch(self, p, bases, offs, bases2=None, offs2=None):
        n = len(offs) - 1
        total = int(offs[-1]) + (int(offs2[-1]) if offs2 is not None else 0)
        cap = max(16, total * 2 + 64)
        out = np.zeros(cap, kmer_dt)
        ql = np.zeros(n, np.int32); ql2 = np.zeros(n, np.int32)
        cnt = self.lib.emu_extract_batch(_ptr(bases), _ptr(offs), _ptr(bases2), _ptr(offs2), C.c_size_t(n),
                                         C.byref(p), _ptr(out), C.c_size_t(cap), _ptr(ql), _ptr(ql2))
        return out[:cnt].copy(), ql, ql2

    def join(self, values, info, tax2species, info_mask, kmer_format, q):
        cap = max(1024, 8 * len(q))
        while True:
            out = np.zeros(cap, match_dt)
            n = self.lib.emu_join(_ptr(values), _ptr(info), C.c_uint64(len(values)), _ptr(tax2species),
                                  C.c_int32(len(tax2species) - 1), C.c_uint32(info_mask), C.c_int(kmer_format),
                                  _ptr(q), C.c_size_t(len(q)), _ptr(out), C.c_size_t(cap))
            if n <= cap:
                return out[:n].copy()
            cap = n

    def join_one_bisection(self, values, info, tax2species, info_mask, kmer_format, q):
        """k_join_dir's search restated sequentially (emu_join_one_bisection) -> (matches, queries that found a target equal to themselves)"""
        self.lib.emu_join_one_bisection.restype = C.c_size_t
        cap = max(1024, 8 * len(q))
        ne = C.c_uint64()
        while True:
            out = np.zeros(cap, match_dt)
            n = self.lib.emu_join_one_bisection(_ptr(values), _ptr(info), C.c_uint64(len(values)), _ptr(tax2species),
                                                C.c_int32(len(tax2species) - 1), C.c_uint32(info_mask), C.c_int(kmer_format),
                                                _ptr(q), C.c_size_t(len(q)), _ptr(out), C.c_size_t(cap), C.byref(ne))
            if n <= cap:
                return out[:n].copy(), int(ne.value)
            cap = n

    def sort_matches(self, m):
        m = m.copy()
        self.lib.emu_sort_matches(_ptr(m), C.c_size_t(len(m)))
        return m

    def score(self, taxarr, p, m, n_reads, ql, ql2):
        canon, parent, depth, under_euk, sp_parent, acc = taxarr
        res = np.zeros(n_reads, result_dt)
        cap = max(1024, len(m) + 16)
        tt = np.zeros(cap, np.int32); tc = np.zeros(cap, np.uint32)
        n = self.lib.emu_score(_ptr(acc), _ptr(canon), _ptr(parent), _ptr(depth), _ptr(under_euk), _ptr(sp_parent), C.c_int32(len(parent) - 1),
                               C.byref(p), _ptr(m), C.c_size_t(len(m)), C.c_size_t(n_reads), _ptr(ql), _ptr(ql2),
                               _ptr(res), _ptr(tt), _ptr(tc), C.c_size_t(cap))
        return res, tt[:n].copy(), tc[:n].copy()


def _emu_score_par(self, taxarr, p, m, n_reads, ql, ql2, presorted=True, use_chain=True):
    canon, parent, depth, under_euk, sp_parent, acc = taxarr
    res = np.zeros(n_reads, result_dt)
    cap = max(1024, len(m) + 16)
    tt = np.zeros(cap, np.int32); tc = np.zeros(cap, np.uint32)
    nchain = C.c_size_t(0)
    n = self.lib.emu_score_par(_ptr(acc), _ptr(canon), _ptr(parent), _ptr(depth), _ptr(under_euk), _ptr(sp_parent), C.c_int32(len(parent) - 1),
                               C.byref(p), _ptr(m), C.c_size_t(len(m)), C.c_size_t(n_reads), _ptr(ql), _ptr(ql2),
                               _ptr(res), _ptr(tt), _ptr(tc), C.c_size_t(cap), C.c_int(1 if presorted else 0),
                               C.c_int(1 if use_chain else 0), C.byref(nchain))
    self.last_chain_reads = nchain.value
    return res, tt[:n].copy(), tc[:n].copy()


Emu.score_par = _emu_score_par


def _emu_load_taxonomy(self, d, taxids):
    """host_db.h: the dense taxonomy arrays libmtb uploads at index-open time."""
    cap = 1 << 22
    arrs = [np.zeros(cap, np.int32) for _ in range(3)]
    under = np.zeros(cap, np.uint8); spp = np.zeros(cap, np.int32); t2s = np.zeros(cap, np.int32); acc = np.zeros(cap, np.uint8)
    mx = C.c_int32()
    ids = np.ascontiguousarray(taxids, dtype=np.int32)
    rc = self.lib.emu_load_taxonomy(d.encode(), _ptr(ids), C.c_size_t(len(ids)), C.c_int32(cap), C.byref(mx), _ptr(arrs[0]), _ptr(arrs[1]),
                                    _ptr(arrs[2]), _ptr(under), _ptr(spp), _ptr(t2s), _ptr(acc))
    assert rc == 0
    n = mx.value + 1
    return (arrs[0][:n].copy(), arrs[1][:n].copy(), arrs[2][:n].copy(), under[:n].copy(), spp[:n].copy(), acc[:n].copy()), t2s[:n].copy()


Emu.load_taxonomy = _emu_load_taxonomy


def tax_arrays(orc: Oracle, tax, world_tax):
    """Dense taxonomy arrays (what libmtb builds at index-open time), derived
    here from the oracle's taxonomy services so the emu path can be tested."""
    L = orc.lib
    mx = L.orc_tax_max_id(tax)
    parent = np.full(mx + 1, -1, np.int32); depth = np.zeros(mx + 1, np.int32)
    under = np.zeros(mx + 1, np.uint8); spp = np.zeros(mx + 1, np.int32)
    euk = [t for t, n in world_tax.name.items() if n == "Eukaryota"]
    euk = euk[0] if euk else 0
    for t in world_tax.parent:
        parent[t] = L.orc_tax_parent(tax, t)
    for t in world_tax.parent:
        d, c = 0, t
        while parent[c] != c:
            c = parent[c]; d += 1
        depth[t] = d
        under[t] = L.orc_tax_is_ancestor(tax, euk, t)
        sp = L.orc_tax_at_rank(tax, t, b"species")
        spp[t] = L.orc_tax_parent(tax, sp) if sp > 0 else 0
    canon = np.where(parent >= 0, np.arange(mx + 1, dtype=np.int32), -1).astype(np.int32)
    acc = np.zeros(mx + 1, np.uint8)
    for t, r in world_tax.rank.items():
        acc[t] = 1 if r in ("", "accession") else 0
    return canon, parent, depth, under, spp, acc


def tax2species_table(orc: Oracle, tax, taxids, mx):
    """KmerMatcher::loadTaxIdList as a dense table."""
    L = orc.lib
    tab = np.zeros(mx + 1, np.int32)
    for t in np.unique(taxids):
        t = int(t)
        sp = L.orc_tax_at_rank(tax, t, b"species")
        c = t
        while c != sp and c > 0:
            tab[c] = sp
            p = L.orc_tax_parent(tax, c)
            if p == c:
                break
            c = p
        if sp > 0:
            tab[sp] = sp
    return tab


def build_toy_db(orc: Oracle, world, p, dbdir, extra=None):
    """Extract target metamers of every genome with the oracle scanners, dedup
    per (value, species), optionally merge `extra` = (values, taxids), write the
    on-disk DB.  Returns (values, taxids)."""
    from metabuli_amd import synth
    per = []
    for tid, g in world.genomes:
        offs = np.array([0, len(g)], dtype=np.uint64)
        pp = default_params(seq_mode=3, syncmer=p.syncmer, smer_len=p.smer_len, kmer_format=p.kmer_format)
        k, _, _ = orc.extract_batch(pp, g, offs)
        per.append(k["value"].copy())
    vals, tids = synth.dedup_targets(world, per)
    if extra is not None:
        ev, et = extra
        vals = np.concatenate([vals, ev]); tids = np.concatenate([tids, et])
        # one entry per (value, taxid); sorted by (value, species==taxid for fillers, taxid)
        sp = np.array([world.tax.species_of(int(t)) for t in tids], dtype=np.int32)
        order = np.lexsort((tids, sp, vals))
        vals, tids, sp = vals[order], tids[order], sp[order]
        keep = np.ones(len(vals), bool)
        keep[1:] = (vals[1:] != vals[:-1]) | (sp[1:] != sp[:-1])
        vals, tids = vals[keep], tids[keep]
    world.tax.write(os.path.join(dbdir, "taxonomy"))
    orc.write_db(dbdir, vals, tids, p)
    return vals, tids
