"""metabuli_amd -- MI355X-native engine for the `metabuli classify` hot path.

This package is plumbing only: it loads the C-ABI shared library
(metabuli_amd/csrc/libmtb.so, declared in include/mtb.h) with ctypes and moves
numpy arrays / device pointers across it.  There is no Python or CPU
implementation of the path: if the HIP library is missing, import-time use
fails loudly.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_CSRC = os.path.join(_HERE, "csrc")
LIB_PATH = os.environ.get("MTB_LIB") or os.path.join(_CSRC, "libmtb.so")   # MTB_LIB: profiling / tuning builds (and, in tests only, the emulated build of tests/hipemu)

kmer_dt = np.dtype([("value", "<u8"), ("qinfo", "<u8")])
match_dt = np.dtype([("qinfo", "<u8"), ("target_id", "<i4"), ("species_id", "<i4"), ("dna", "<u4"),
                     ("reh", "<u2"), ("ham", "u1"), ("pad", "u1")])
result_dt = np.dtype([("classification", "<i4"), ("score", "<f4"), ("qlen", "<i4"), ("qlen2", "<i4"),
                      ("is_classified", "u1"), ("flag", "u1"), ("n_taxcnt", "<u2"), ("taxcnt_off", "<u4")])

MTB_OK, MTB_ERR_ARG, MTB_ERR_IO, MTB_ERR_DEVICE, MTB_ERR_CAPACITY, MTB_ERR_OOM, MTB_ERR_UNSUPPORTED = range(7)


class Params(C.Structure):
    _fields_ = [("seq_mode", C.c_int32), ("syncmer", C.c_int32), ("smer_len", C.c_int32),
                ("kmer_format", C.c_int32), ("min_cons_cnt", C.c_int32), ("min_cons_cnt_euk", C.c_int32),
                ("min_score", C.c_float), ("min_sp_score", C.c_float), ("tie_ratio", C.c_float),
                ("accession_level", C.c_int32), ("skip_redundancy", C.c_int32)]


class BatchStats(C.Structure):
    _fields_ = [("ms_extract", C.c_float), ("ms_sort", C.c_float), ("ms_join", C.c_float), ("ms_regroup", C.c_float),
                ("ms_segsort", C.c_float), ("ms_score", C.c_float), ("ms_total", C.c_float),
                ("n_reads", C.c_uint64), ("n_bases", C.c_uint64), ("n_kmers", C.c_uint64),
                ("n_matches", C.c_uint64), ("n_targets", C.c_uint64),
                ("ms_kernel", C.c_float * 11), ("n_launch", C.c_uint32 * 11), ("n_generic_reads", C.c_uint64), ("n_slot_reads", C.c_uint64),
                ("n_deferred_reads", C.c_uint64), ("n_many_reads", C.c_uint64), ("n_many_matches", C.c_uint64), ("n_many_kept", C.c_uint64),
                ("join_variant", C.c_int32), ("join_tuned", C.c_int32), ("join_tune_ms", C.c_float * 3),
                ("join_tiles", C.c_uint32), ("join_tiles_windowed", C.c_uint32), ("join_tiles_outside", C.c_uint32)]


SHARE_BYTES = 4 * 64 + 4 * 8 + 8 + 2 * 4 + 5 * 4 + 4 + 8 + 4 * 8      # sizeof(mtb_index_share) (tests/test_abi.py checks it against the header)
JOIN_VARIANTS = {0: "other", 1: "q1w6", 2: "q2w5", 3: "window", -3: "q1w5", -4: "q2w6", -15: "windoww5", -16: "windoww6", -17: "windoww7"}      # mtb_batch_stats.join_variant (mtb_join_variant)


class JoinFootprint(C.Structure):
    _fields_ = [("n_queries", C.c_uint64), ("distinct_buckets", C.c_uint64), ("dir_sectors", C.c_uint64),
                ("target_sectors", C.c_uint64), ("n_buckets", C.c_uint64), ("n_targets", C.c_uint64)]


KERNEL_NAMES = ["extract_count", "extract_emit", "radix_hist", "radix_scatter", "join", "regroup", "segsort", "score", "scan", "score_fast", "score_many"]


class MtbError(RuntimeError):
    def __init__(self, status, msg):
        super().__init__(f"mtb status {status}: {msg}")
        self.status = status


def build(force=False):
    """Compile libmtb.so for gfx950 with hipcc (cross-compiles without a GPU)."""
    if force and os.path.exists(LIB_PATH):
        os.remove(LIB_PATH)
    subprocess.check_call(["make", "-C", _CSRC, "libmtb.so"], stdout=subprocess.DEVNULL)
    return LIB_PATH


_lib = None


def lib():
    """Load the HIP library.  No fallback: raises if it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(f"{LIB_PATH} not found: the HIP extension must be built (python -c 'import __graft_entry__ as g; g.build()'); "
                           "there is no CPU implementation of this path")
    L = C.CDLL(LIB_PATH)
    L.mtb_version.restype = C.c_char_p
    L.mtb_last_error.restype = C.c_char_p
    L.mtb_index_num_targets.restype = C.c_uint64
    L.mtb_index_num_targets.argtypes = [C.c_void_p]
    for f in ("mtb_tax_lca", "mtb_tax_species", "mtb_tax_parent", "mtb_tax_max_id"):
        getattr(L, f).restype = C.c_int32
    L.mtb_tax_lca.argtypes = [C.c_void_p, C.c_int32, C.c_int32]
    L.mtb_tax_species.argtypes = [C.c_void_p, C.c_int32]
    L.mtb_tax_parent.argtypes = [C.c_void_p, C.c_int32]
    L.mtb_tax_max_id.argtypes = [C.c_void_p]
    L.mtb_tax_original_id.restype = C.c_int32
    L.mtb_tax_original_id.argtypes = [C.c_void_p, C.c_int32]
    for f in ("mtb_tax_rank", "mtb_tax_name"):
        getattr(L, f).restype = C.c_char_p
        getattr(L, f).argtypes = [C.c_void_p, C.c_int32]
    _lib = L
    return L


def part_bounds(dbdir, n_parts):
    """lower amino-acid-part bound of each of n_parts value ranges of a database directory (mtb_index_part_bounds)"""
    b = np.zeros(n_parts, np.uint64)
    _chk(lib().mtb_index_part_bounds(dbdir.encode(), C.c_uint32(n_parts), _p(b)))
    return b


def default_params(**kw):
    """classify defaults for a syncmer DB written by `build` (db.parameters overrides apply at index open)."""
    p = Params(seq_mode=1, syncmer=1, smer_len=5, kmer_format=2, min_cons_cnt=4, min_cons_cnt_euk=9,
               min_score=0.0, min_sp_score=0.0, tie_ratio=0.95, accession_level=0, skip_redundancy=1)
    for k, v in kw.items():
        setattr(p, k, v)
    return p


def _p(a):
    if a is None:
        return None
    if isinstance(a, int):
        return C.c_void_p(a)
    return a.ctypes.data_as(C.c_void_p)


def _chk(st):
    if st != MTB_OK:
        raise MtbError(st, lib().mtb_last_error().decode())


def compact_taxcnt(res, tt, tc):
    """The C ABI leaves every read's taxID:match_count entries at res.taxcnt_off
    inside bound-sized slots; pack them back to back (host-side convenience)."""
    n = res["n_taxcnt"].astype(np.int64)
    new_off = np.zeros(len(res) + 1, np.int64)
    np.cumsum(n, out=new_off[1:])
    idx = np.repeat(res["taxcnt_off"].astype(np.int64) - new_off[:-1], n) + np.arange(int(new_off[-1]))
    out = res.copy()
    out["taxcnt_off"] = new_off[:-1].astype(np.uint32)
    return out, tt[idx].copy(), tc[idx].copy()


class Context:
    def __init__(self, device=0, stream=0):
        self.L = lib()
        h = C.c_void_p()
        _chk(self.L.mtb_ctx_create(C.c_int(device), C.c_void_p(stream), C.byref(h)))
        self.h = h

    def close(self):
        if self.h:
            self.L.mtb_ctx_destroy(self.h)
            self.h = None

    def sync(self):
        _chk(self.L.mtb_ctx_sync(self.h))

    def set_profiling(self, on):
        _chk(self.L.mtb_ctx_set_profiling(self.h, C.c_int(1 if on else 0)))

    def set_placement_probe(self, on):
        """slot buffers >= 8 GB are chosen among candidate allocations (for processes with a long device-allocation history of their own)"""
        _chk(self.L.mtb_ctx_set_placement_probe(self.h, C.c_int(1 if on else 0)))

    def set_workspace_limit(self, nbytes):
        """workspace budget of a batch in bytes (0 = automatic from hipMemGetInfo): forces sub-batches in tests"""
        _chk(self.L.mtb_ctx_set_workspace_limit(self.h, C.c_uint64(int(nbytes))))

    def set_join_variant(self, name):
        """pin the short-read join's instantiation ("q1w6", "q2w5", "window") or hand the choice back to the context's tuner ("auto")"""
        _chk(self.L.mtb_ctx_set_join_variant(self.h, C.c_int({"auto": 0, "q1w6": 1, "q2w5": 2, "window": 3}[name])))

    def set_option(self, name, value):
        """one of the library's experiment switches on this live context (value None = unset); the environment is read once, at creation"""
        _chk(self.L.mtb_ctx_set_option(self.h, name.encode(), None if value is None else str(value).encode()))

    @property
    def last_sub_batches(self):
        self.L.mtb_ctx_last_sub_batches.restype = C.c_uint32
        return int(self.L.mtb_ctx_last_sub_batches(self.h))

    @property
    def last_scratch_bytes(self):
        self.L.mtb_ctx_last_scratch_bytes.restype = C.c_uint64
        return int(self.L.mtb_ctx_last_scratch_bytes(self.h))

    def set_streams(self, n):
        _chk(self.L.mtb_ctx_set_streams(self.h, C.c_int(n)))

    # ---- index ----
    def open_index(self, dbdir, params, taxonomy_dir=None):
        h = C.c_void_p()
        _chk(self.L.mtb_index_open(self.h, dbdir.encode(), taxonomy_dir.encode() if taxonomy_dir else None,
                                   C.byref(params), C.byref(h)))
        return Index(self, h)

    def clone_index(self, src):
        """a copy of a resident index (of any context) on this context's GPU: peer copies, no file access (mtb_index_clone)"""
        h = C.c_void_p()
        _chk(self.L.mtb_index_clone(src.h, self.h, C.byref(h)))
        return Index(self, h)

    def import_index(self, share, taxonomy_dir, taxid_list, params):
        """an index another PROCESS of the node holds resident (Index.export() there, the bytes sent over any channel) copied to this context's GPU
        device to device (mtb_index_import); the exporter must keep its index open and idle until this returns"""
        h = C.c_void_p()
        tl = np.ascontiguousarray(taxid_list, dtype=np.int32)
        buf = C.create_string_buffer(bytes(share), SHARE_BYTES)
        _chk(self.L.mtb_index_import(self.h, buf, taxonomy_dir.encode(), _p(tl), C.c_size_t(len(tl)), C.byref(params), C.byref(h)))
        return Index(self, h)

    def index_from_device(self, d_values, d_info, n_targets, taxonomy_dir, taxid_list, params):
        h = C.c_void_p()
        tl = np.ascontiguousarray(taxid_list, dtype=np.int32)
        _chk(self.L.mtb_index_from_device(self.h, C.c_void_p(d_values), C.c_void_p(d_info), C.c_uint64(n_targets),
                                          taxonomy_dir.encode(), _p(tl), C.c_size_t(len(tl)), C.byref(params), C.byref(h)))
        return Index(self, h)

    def synth_index(self, seed, n_filler, tax_lo, tax_hi, real_values, real_taxids, d_values, d_info):
        rv = np.ascontiguousarray(real_values, dtype=np.uint64)
        rt = np.ascontiguousarray(real_taxids, dtype=np.int32)
        n = C.c_uint64()
        _chk(self.L.mtb_synth_index(self.h, C.c_uint64(seed), C.c_uint64(n_filler), C.c_int32(tax_lo), C.c_int32(tax_hi),
                                    _p(rv), _p(rt), C.c_uint64(len(rv)), C.c_void_p(d_values), C.c_void_p(d_info), C.byref(n)))
        return n.value

    # ---- stages (host buffers) ----
    def extract(self, params, bases, offs, bases2=None, offs2=None, cap=None):
        n = len(offs) - 1
        total = int(offs[-1]) + (int(offs2[-1]) if offs2 is not None else 0)
        cap = cap or max(16, 2 * total + 64)
        out = np.zeros(cap, kmer_dt)
        ql = np.zeros(n, np.int32); ql2 = np.zeros(n, np.int32)
        cnt = C.c_uint64()
        _chk(self.L.mtb_extract(self.h, C.byref(params), _p(bases), _p(offs), _p(bases2), _p(offs2), C.c_uint64(n),
                                _p(out), C.c_uint64(cap), C.byref(cnt), _p(ql), _p(ql2)))
        return out[:cnt.value].copy(), ql, ql2

    def sort_kmers(self, kmers):
        k = np.ascontiguousarray(kmers).copy()
        _chk(self.L.mtb_sort_kmers(self.h, _p(k), C.c_uint64(len(k))))
        return k

    def match(self, index, sorted_kmers, cap=None):
        cap = cap or max(1024, 4 * len(sorted_kmers))
        while True:
            out = np.zeros(cap, match_dt)
            cnt = C.c_uint64()
            st = self.L.mtb_match_kmers(self.h, index.h, _p(sorted_kmers), C.c_uint64(len(sorted_kmers)), _p(out),
                                        C.c_uint64(cap), C.byref(cnt))
            if st == MTB_ERR_CAPACITY:
                cap = cnt.value + 16
                continue
            _chk(st)
            return out[:cnt.value].copy()

    def sort_matches(self, matches, n_reads):
        m = np.ascontiguousarray(matches).copy()
        _chk(self.L.mtb_sort_matches(self.h, _p(m), C.c_uint64(len(m)), C.c_uint64(n_reads)))
        return m

    def score(self, index, params, sorted_matches, n_reads, qlen, qlen2):
        res = np.zeros(n_reads, result_dt)
        cap = max(1024, len(sorted_matches) + 16)
        tt = np.zeros(cap, np.int32); tc = np.zeros(cap, np.uint32)
        n = C.c_uint64()
        _chk(self.L.mtb_score(self.h, index.h, C.byref(params), _p(sorted_matches), C.c_uint64(len(sorted_matches)),
                              C.c_uint64(n_reads), _p(qlen), _p(qlen2), _p(res), _p(tt), _p(tc), C.c_uint64(cap), C.byref(n)))
        return compact_taxcnt(res, tt, tc)

    # ---- fused batch ----
    def classify_batch(self, index, params, bases, offs, bases2=None, offs2=None, taxcnt_cap=None):
        """taxcnt_cap: capacity of the host arrays for the packed taxID:count lists (default: ample); too small -> the call is
        repeated with the size the library reports (self.last_capacity_retries counts those)"""
        n = len(offs) - 1
        res = np.zeros(n, result_dt)
        cap = max(1024, 64 * n) if taxcnt_cap is None else int(taxcnt_cap)
        self.last_capacity_retries = 0
        while True:
            tt = np.zeros(cap, np.int32); tc = np.zeros(cap, np.uint32)
            cnt = C.c_uint64()
            st = self.L.mtb_classify_batch(self.h, index.h, C.byref(params), _p(bases), _p(offs), _p(bases2), _p(offs2),
                                           C.c_uint64(n), _p(res), _p(tt), _p(tc), C.c_uint64(cap), C.byref(cnt))
            if st == MTB_ERR_CAPACITY and cnt.value > cap:
                cap = cnt.value
                self.last_capacity_retries += 1
                continue
            _chk(st)
            return compact_taxcnt(res, tt, tc)

    @staticmethod
    def pack_reads(bases, offs):
        """host-side 2-bit form of a read batch as mtb_classify_batch_packed takes it (what the C++ parser writes): per read
        ceil(len / 8) groups of 8 bases -- 16 bits of codes (A 0, C 1, T 2, G 3 and their IUPAC classes, mtb_core.h; base j of a
        group at bits 2j, little endian) + 8 bits that mark the bases outside those classes -> (packed2, nmask, lens)"""
        cls = np.full(256, 255, np.uint8)
        for c, letters in enumerate(("ARW", "CMS", "HTY", "BDGKU")):
            for ch in letters:
                cls[ord(ch)] = c; cls[ord(ch.lower())] = c
        bases = np.ascontiguousarray(bases, dtype=np.uint8)
        offs = np.asarray(offs, dtype=np.uint64)
        lens = (offs[1:] - offs[:-1]).astype(np.uint32)
        groups = (lens.astype(np.int64) + 7) // 8
        gstart = np.zeros(len(lens) + 1, np.int64); np.cumsum(groups, out=gstart[1:])
        code = cls[bases]
        # position of every base inside the padded (8 per group) layout
        read_of = np.repeat(np.arange(len(lens)), lens.astype(np.int64))
        k = np.arange(len(bases), dtype=np.int64) - offs[:-1].astype(np.int64)[read_of]
        slot = gstart[:-1][read_of] * 8 + k
        padded_code = np.zeros(int(gstart[-1]) * 8, np.uint16)
        padded_bad = np.zeros(int(gstart[-1]) * 8, np.uint8)
        padded_code[slot] = np.where(code < 4, code, 0)
        padded_bad[slot] = code >= 4
        sh2 = (2 * np.arange(8)).astype(np.uint16)
        packed = (padded_code.reshape(-1, 8) << sh2).sum(axis=1).astype(np.uint16)
        nmask = (padded_bad.reshape(-1, 8).astype(np.uint16) << np.arange(8).astype(np.uint16)).sum(axis=1).astype(np.uint8)
        return packed.view(np.uint8).copy(), nmask, lens

    def classify_batch_packed(self, index, params, bases, offs, bases2=None, offs2=None):
        """mtb_classify_batch_packed on reads packed here (pack_reads); results as classify_batch"""
        n = len(offs) - 1
        p1 = self.pack_reads(bases, offs)
        p2 = self.pack_reads(bases2, offs2) if bases2 is not None else (None, None, None)
        res = np.zeros(n, result_dt)
        cap = max(1024, 8 * n)
        while True:
            tt = np.zeros(cap, np.int32); tc = np.zeros(cap, np.uint32)
            cnt = C.c_uint64()
            st = self.L.mtb_classify_batch_packed(self.h, index.h, C.byref(params), _p(p1[0]), _p(p1[1]), _p(p1[2]), _p(p2[0]), _p(p2[1]), _p(p2[2]),
                                                  C.c_uint64(n), _p(res), _p(tt), _p(tc), C.c_uint64(cap), C.byref(cnt))
            if st == MTB_ERR_CAPACITY and cnt.value > cap:
                cap = cnt.value
                continue
            _chk(st)
            return compact_taxcnt(res, tt, tc)

    def classify_batches_packed_async(self, index, params, batches):
        """mtb_classify_batch_packed_async over a list of (bases, offs, bases2, offs2) batches the way a driver uses it: batch k's results are
        taken after the call for batch k + 1 has returned (the last one's after mtb_ctx_wait_results) -> list of (results, taxcnt_tax, taxcnt_cnt)"""
        out, held = [], None
        for (bases, offs, bases2, offs2) in batches:
            n = len(offs) - 1
            p1 = self.pack_reads(bases, offs)
            p2 = self.pack_reads(bases2, offs2) if bases2 is not None else (None, None, None)
            res = np.frombuffer(bytearray(b"\xEE" * (n * result_dt.itemsize)), dtype=result_dt)
            cap = max(1024, 3 * n)
            while True:
                tt = np.full(cap, -286331154, np.int32); tc = np.full(cap, 0xEEEEEEEE, np.uint32)
                cnt = C.c_uint64()
                st = self.L.mtb_classify_batch_packed_async(self.h, index.h, C.byref(params), _p(p1[0]), _p(p1[1]), _p(p1[2]), _p(p2[0]), _p(p2[1]), _p(p2[2]),
                                                            C.c_uint64(n), _p(res), _p(tt), _p(tc), C.c_uint64(cap), C.byref(cnt))
                if st == MTB_ERR_CAPACITY and cnt.value > cap:
                    cap = cnt.value
                    continue
                _chk(st)
                break
            if held is not None:                       # the previous batch's arrays are complete now
                out.append(compact_taxcnt(*held))
            held = (res, tt, tc, p1, p2)[:3]
        _chk(self.L.mtb_ctx_wait_results(self.h))
        if held is not None:
            out.append(compact_taxcnt(*held))
        return out

    def reserve(self, params, n_reads, n_bases):
        """mtb_ctx_reserve: grow the workspace of a short-read batch of that size ahead of time"""
        _chk(self.L.mtb_ctx_reserve(self.h, C.byref(params), C.c_uint64(n_reads), C.c_uint64(n_bases)))

    def prefetch_stats(self):
        """(prefetches issued, prefetches a classify call used instead of uploading its batch again) -- mtb_ctx_prefetch_stats"""
        a, b = C.c_uint64(), C.c_uint64()
        _chk(self.L.mtb_ctx_prefetch_stats(self.h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def classify_batches_packed_prefetched(self, index, params, batches):
        """The driver's protocol (mtb.h): prefetch(k+1), classify(k), prefetch(k+2), classify(k+1) ... over a list of
        (bases, offs, bases2, offs2) batches -> list of (results, taxcnt_tax, taxcnt_cnt).  The packed arrays stay alive until the end."""
        packed = []
        for (bases, offs, bases2, offs2) in batches:
            packed.append((len(offs) - 1, self.pack_reads(bases, offs), self.pack_reads(bases2, offs2) if bases2 is not None else (None, None, None)))
        out = []
        for k, (n, p1, p2) in enumerate(packed):
            if k + 1 < len(packed):
                n2, q1, q2 = packed[k + 1]
                _chk(self.L.mtb_prefetch_batch_packed(self.h, C.byref(params), _p(q1[0]), _p(q1[1]), _p(q1[2]), _p(q2[0]), _p(q2[1]), _p(q2[2]), C.c_uint64(n2)))
            res = np.zeros(n, result_dt)
            cap = max(1024, 8 * n)
            while True:
                tt = np.zeros(cap, np.int32); tc = np.zeros(cap, np.uint32)
                cnt = C.c_uint64()
                st = self.L.mtb_classify_batch_packed(self.h, index.h, C.byref(params), _p(p1[0]), _p(p1[1]), _p(p1[2]), _p(p2[0]), _p(p2[1]), _p(p2[2]),
                                                      C.c_uint64(n), _p(res), _p(tt), _p(tc), C.c_uint64(cap), C.byref(cnt))
                if st == MTB_ERR_CAPACITY and cnt.value > cap:
                    cap = cnt.value
                    continue
                _chk(st)
                break
            out.append(compact_taxcnt(res, tt, tc))
        return out

    def classify_batch_device(self, index, params, d_bases, d_offs, d_bases2, d_offs2, n_reads, n_bases,
                              d_results, d_tc_tax, d_tc_cnt, tc_cap):
        cnt = C.c_uint64()
        _chk(self.L.mtb_classify_batch_device(self.h, index.h, C.byref(params), C.c_void_p(d_bases), C.c_void_p(d_offs),
                                              C.c_void_p(d_bases2) if d_bases2 else None,
                                              C.c_void_p(d_offs2) if d_offs2 else None,
                                              C.c_uint64(n_reads), C.c_uint64(n_bases), C.c_void_p(d_results),
                                              C.c_void_p(d_tc_tax), C.c_void_p(d_tc_cnt), C.c_uint64(tc_cap), C.byref(cnt)))
        return cnt.value

    # ---- partitioned index: device-buffer stage calls ----
    def open_index_part(self, dbdir, params, part, n_parts, taxonomy_dir=None):
        h = C.c_void_p()
        _chk(self.L.mtb_index_open_part(self.h, dbdir.encode(), taxonomy_dir.encode() if taxonomy_dir else None,
                                        C.byref(params), C.c_uint32(part), C.c_uint32(n_parts), C.byref(h)))
        return Index(self, h)

    def part_extract(self, params, d_bases, d_offs, d_bases2, d_offs2, n_reads, bounds, overlapping=True):
        """-> (device pointer of the sorted metamers (owned by the context), n_kmers, counts per partition, starts per partition).
        overlapping=False: the legacy consecutive runs (full sort on bits [24, 64), exact-segment scoring at home)."""
        b = np.ascontiguousarray(bounds, dtype=np.uint64)
        ptr = C.c_void_p(); nk = C.c_uint64()
        counts = np.zeros(len(b), np.uint64); starts = np.zeros(len(b), np.uint64)
        _chk(self.L.mtb_part_extract(self.h, C.byref(params), C.c_void_p(d_bases), C.c_void_p(d_offs),
                                     C.c_void_p(d_bases2) if d_bases2 else None, C.c_void_p(d_offs2) if d_offs2 else None,
                                     C.c_uint64(n_reads), _p(b), C.c_uint32(len(b)), C.byref(ptr), C.byref(nk), _p(counts),
                                     _p(starts) if overlapping else None))
        if not overlapping:
            starts[1:] = np.cumsum(counts)[:-1]
        return ptr.value or 0, nk.value, counts, starts

    def part_join(self, index, d_kmers, n, d_out, cap):
        """-> (status, count); status MTB_ERR_CAPACITY means `count` entries are needed"""
        cnt = C.c_uint64()
        st = self.L.mtb_part_join(self.h, index.h, C.c_void_p(d_kmers), C.c_uint64(n), C.c_void_p(d_out), C.c_uint64(cap), C.byref(cnt))
        if st != MTB_ERR_CAPACITY:
            _chk(st)
        return st, cnt.value

    def part_score(self, index, params, d_matches, n_matches, n_reads):
        res = np.empty(n_reads, result_dt)
        cap = max(1024, 24 * n_reads + 4096)        # one slot per position bucket (18 for a 150 bp read); more -> exact retry
        while True:
            tt = np.empty(cap, np.int32); tc = np.empty(cap, np.uint32)      # (never zero-filled: 64 slots x 8 bytes x 2 M reads were 1 GB of memset per call)
            n = C.c_uint64()
            st = self.L.mtb_part_score(self.h, index.h, C.byref(params), C.c_void_p(d_matches), C.c_uint64(n_matches),
                                       C.c_uint64(n_reads), _p(res), _p(tt), _p(tc), C.c_uint64(cap), C.byref(n))
            if st == MTB_ERR_CAPACITY and n.value > cap:
                cap = n.value
                continue
            _chk(st)
            return res, tt[:n.value], tc[:n.value]          # the library packs the lists on the device (taxcnt_off = running total)

    def join_footprint(self, index):
        """index-side working set of the last fused batch's directory join (mtb_ctx_join_footprint; diagnostic)"""
        f = JoinFootprint()
        _chk(self.L.mtb_ctx_join_footprint(self.h, index.h, C.byref(f)))
        return f

    def join_run_histogram(self, index):
        """lengths of the candidate runs the last fused batch's queries met (mtb_ctx_join_run_histogram; diagnostic) ->
        dict(no_candidate, queries_by_log2[32], candidates_by_log2[24])"""
        h = np.zeros(64, np.uint64)
        _chk(self.L.mtb_ctx_join_run_histogram(self.h, index.h, _p(h)))
        return dict(no_candidate=int(h[0]), queries_by_log2=[int(x) for x in h[1:33]], candidates_by_log2=[int(x) for x in h[40:64]],
                    exact_queries=int(h[33]), exact_run_targets=int(h[34]))

    def last_stats(self):
        s = BatchStats()
        _chk(self.L.mtb_last_batch_stats(self.h, C.byref(s)))
        return s


class Index:
    def __init__(self, ctx, h):
        self.ctx = ctx
        self.h = h

    @property
    def num_targets(self):
        return self.ctx.L.mtb_index_num_targets(self.h)

    def open_stats(self):
        """mtb_index_open_stats: how the database files came in (chunked decode)"""
        o = np.zeros(4, np.uint64)
        _chk(self.ctx.L.mtb_index_open_stats(self.h, _p(o)))
        return dict(chunks=int(o[0]), chunk_words=int(o[1]), peak_bytes=int(o[2]), packed_on_load=bool(o[3]))

    def export(self):
        """mtb_index_export: the record (bytes) another process of the node hands to Context.import_index; keep this index open and idle meanwhile"""
        buf = C.create_string_buffer(SHARE_BYTES)
        _chk(self.ctx.L.mtb_index_export(self.h, buf))
        return buf.raw

    def seal(self):
        """mtb_index_seal: packed state + info[] released (the lender of a borrowed info array may free it afterwards)"""
        _chk(self.ctx.L.mtb_index_seal(self.h))

    def state(self):
        """directory depth (0 = none), packed / sealed flags of the resident target array (mtb_index_state)"""
        d = C.c_int32(); pk = C.c_int32(); sl = C.c_int32()
        _chk(self.ctx.L.mtb_index_state(self.h, C.byref(d), C.byref(pk), C.byref(sl)))
        return dict(dir_depth=d.value, packed=bool(pk.value), sealed=bool(sl.value))

    def original_id(self, taxid):
        """TaxonomyWrapper::getOriginalTaxID: internal id (what results carry) -> the id the reports print"""
        return int(self.ctx.L.mtb_tax_original_id(self.h, C.c_int32(int(taxid))))

    def slice(self, lo_value, hi_value, is_last):
        """device view of the value range [lo, hi) (mtb_index_slice); close it before the parent"""
        h = C.c_void_p()
        _chk(self.ctx.L.mtb_index_slice(self.h, C.c_uint64(int(lo_value)), C.c_uint64(int(hi_value)), C.c_int(1 if is_last else 0), C.byref(h)))
        return Index(self.ctx, h)

    def run_histogram(self):
        """candidate-run lengths over the whole index (mtb_index_run_histogram; brings the index to the flat state) ->
        (runs_by_log2[32], targets_by_log2[32])"""
        h = np.zeros(64, np.uint64)
        _chk(self.ctx.L.mtb_index_run_histogram(self.h, _p(h)))
        return [int(x) for x in h[:32]], [int(x) for x in h[32:]]

    def write(self, dbdir, split_num=4096):
        """the index in the reference's on-disk format (mtb_index_write); dbdir must exist"""
        _chk(self.ctx.L.mtb_index_write(self.h, dbdir.encode(), C.c_int(split_num)))

    def download(self):
        n = self.num_targets
        v = np.zeros(n, np.uint64); i = np.zeros(n, np.uint32)
        _chk(self.ctx.L.mtb_index_download(self.h, _p(v), _p(i), C.c_uint64(n)))
        return v, i

    def close(self):
        if self.h:
            self.ctx.L.mtb_index_close(self.h)
            self.h = None
