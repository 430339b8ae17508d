"""Multi-GPU decomposition of the classify hot path.

Row 1 (index fits one GPU, the default):

Reads are independent units: with the index replicated on every GPU the path
shards by contiguous read ranges and needs NO data-path collective.  The only
communication is control-plane: a barrier around timed regions and the sum of
the per-taxon read counts (Classifier.cpp:201-203) at the end.  One process
per GPU; `torch.distributed` backend "nccl" (= RCCL) on GPUs, "gloo" in the CPU
tests.

Row 2 (index larger than one HBM): `classify_partitioned` below.  The flat
target array is range-partitioned by value at amino-acid-part boundaries, one
range per GPU; per batch the sorted query metamers travel to the owner of their
range (all-to-all #1, 16-byte records), are joined there, and the 24-byte
matches travel back to the read's home GPU (all-to-all #2), which places them into
its per-read slot segments by the metamer's ordinal and runs the slot scorers -- the kernels of the
replicated path (k_join_dir, k_score_fast, k_score) on both sides.  Two exchange steps, no all-reduce on the data path.
"""
from __future__ import annotations

import numpy as np


def shard_range(n_items: int, rank: int, world_size: int):
    """Contiguous, balanced [lo, hi) of rank `rank` (first n % w ranks get one extra)."""
    q, r = divmod(n_items, world_size)
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


def shard_reads(bases: np.ndarray, offs: np.ndarray, rank: int, world_size: int):
    """Slice a concatenated read batch to this rank's reads; offsets rebased to 0."""
    lo, hi = shard_range(len(offs) - 1, rank, world_size)
    o = offs[lo:hi + 1]
    return bases[int(o[0]):int(o[-1])], (o - o[0]).astype(np.uint64), lo, hi


def tax_counts(classification: np.ndarray, max_taxid: int) -> np.ndarray:
    """Dense per-taxon read counts of one shard (taxCounts[classification]++)."""
    return np.bincount(np.asarray(classification, dtype=np.int64), minlength=max_taxid + 1).astype(np.int64)


def allreduce_tax_counts(counts: np.ndarray, dist=None) -> np.ndarray:
    """Sum the per-shard count vectors over all ranks (a few MB at most)."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return counts
    import torch
    t = torch.from_numpy(counts.copy())
    if dist.get_backend() == "nccl":
        t = t.cuda()
    dist.all_reduce(t)
    return t.cpu().numpy()


# --------------------------------------------------------------------------
# Row 2: range-partitioned index, two all-to-all exchanges per batch
# --------------------------------------------------------------------------
class GpuStages:
    """The three device stages of one rank (libmtb's mtb_part_* calls); tensors are int64 views of the
    16-byte metamer records ([n, 2]) and 24-byte match records ([n, 3]) on this rank's GPU."""

    def __init__(self, ctx, index, params, device):
        import torch
        self.ctx, self.index, self.params, self.torch = ctx, index, params, torch
        self.device = torch.device(device)
        self.n_reads = 0
        self.overlapping = True     # the product path: ordinal tags, prefix-granular (overlapping) runs, slot scoring at home

    def set_reads(self, d_bases, d_offs, n_reads, d_bases2=None, d_offs2=None):
        """device tensors: bases uint8, offs int64 (n_reads + 1); mates for seq_mode 2"""
        self.reads = (d_bases, d_offs, d_bases2, d_offs2)
        self.n_reads = n_reads

    def _fence(self):
        # the library runs on its own HIP stream and synchronises before it returns; work torch has queued
        # (collectives, cat/copies) must have finished before the library reads those tensors
        self.torch.cuda.current_stream(self.device).synchronize()

    def extract_sorted(self, bounds):
        torch = self.torch
        b, o, b2, o2 = self.reads
        self._fence()
        ptr, nk, counts, starts = self.ctx.part_extract(self.params, b.data_ptr(), o.data_ptr(), b2.data_ptr() if b2 is not None else 0,
                                                        o2.data_ptr() if o2 is not None else 0, self.n_reads, bounds, overlapping=self.overlapping)
        if nk == 0:
            return torch.empty((0, 2), dtype=torch.int64, device=self.device), [0] * len(counts), [0] * len(counts)

        class _View:            # zero-copy view of the context-owned buffer (valid until the next stage call)
            __cuda_array_interface__ = {"shape": (int(nk), 2), "typestr": "<i8", "data": (int(ptr), False), "version": 2}
        return torch.as_tensor(_View(), device=self.device), [int(c) for c in counts], [int(x) for x in starts]

    def join(self, run):
        torch = self.torch
        n = int(run.shape[0])
        if n == 0:
            return torch.empty((0, 3), dtype=torch.int64, device=self.device)
        run = run.contiguous()
        self._fence()
        cap = n + n // 4 + 1024
        while True:
            out = torch.empty((cap, 3), dtype=torch.int64, device=self.device)
            st, cnt = self.ctx.part_join(self.index, run.data_ptr(), n, out.data_ptr(), cap)
            if st == 0:
                return out[:cnt]
            cap = cnt + 16

    def score(self, matches):
        matches = matches.contiguous()
        self._fence()
        return self.ctx.part_score(self.index, self.params, matches.data_ptr(), int(matches.shape[0]), self.n_reads)


# One message is bounded: torch 2.10 + RCCL 2.26 returned corrupt data for variable-split all_to_all_single calls beyond
# ~1 GiB per call (measured on MI355X at world_size 1 only -- no multi-GPU node was available to the builder: 1.0 GiB
# intact, 1.5 GiB not; to be re-tested at world_size >= 2).  512 MiB per peer and round is far above the xGMI latency regime.
_XCHG_BYTES = 512 << 20


def _exchange(torch, dist, send, send_counts, width, xdev, send_starts=None):
    """all-to-all(v) of [n, width] int64 rows; returns (received rows, per-source row counts).

    One small collective for the counts: every rank all-gathers its send-count vector, so each rank reads its receive
    counts (a column of the matrix) and the global maximum (number of rounds) from ONE host copy -- no second collective,
    no second sync.  The payload moves as point-to-point messages between views of the send / receive buffers
    (batch_isend_irecv = one ncclGroup per round on RCCL): nothing is concatenated or copied back, the own share is a
    device-to-device copy, and a peer's share is cut into rounds of at most _XCHG_BYTES."""
    world, me = dist.get_world_size(), dist.get_rank()
    if world == 1:
        # one rank owns everything: its share is already where it has to be -- a view, no collective, no copy
        n0 = int(send_counts[0]); o0 = int(send_starts[0]) if send_starts is not None else 0
        return send[o0:o0 + n0], [n0]
    sc = torch.tensor([int(x) for x in send_counts], dtype=torch.int64, device=xdev)
    allc = torch.empty(world * world, dtype=torch.int64, device=xdev)
    dist.all_gather_into_tensor(allc, sc)
    mat = allc.cpu().view(world, world)                       # mat[src][dst]
    recv_counts = [int(x) for x in mat[:, me].tolist()]
    send_counts = [int(x) for x in send_counts]
    send = send.to(xdev).contiguous()
    recv = torch.empty((sum(recv_counts), width), dtype=torch.int64, device=xdev)
    # the runs of a sender are consecutive unless it says otherwise (prefix-granular cuts let neighbouring runs overlap)
    s_off = np.asarray(send_starts, dtype=np.int64) if send_starts is not None else np.concatenate([[0], np.cumsum(send_counts)]).astype(np.int64)
    r_off = np.concatenate([[0], np.cumsum(recv_counts)]).astype(np.int64)
    if send_counts[me]:
        recv[int(r_off[me]): int(r_off[me]) + send_counts[me]] = send[int(s_off[me]): int(s_off[me]) + send_counts[me]]
    chunk = max(1, _XCHG_BYTES // (8 * width))                 # rows per peer per round
    mx = int(mat.max().item()) if world > 1 else 0
    for k in range((mx + chunk - 1) // chunk):
        ops = []
        for d in range(1, world):                              # peer order staggered by rank: no hot spot on the point-to-point links
            to, frm = (me + d) % world, (me - d) % world
            s_n = max(0, min(chunk, send_counts[to] - k * chunk))
            r_n = max(0, min(chunk, recv_counts[frm] - k * chunk))
            if s_n:
                ops.append(dist.P2POp(dist.isend, send[int(s_off[to]) + k * chunk: int(s_off[to]) + k * chunk + s_n], to))
            if r_n:
                ops.append(dist.P2POp(dist.irecv, recv[int(r_off[frm]) + k * chunk: int(r_off[frm]) + k * chunk + r_n], frm))
        if ops:
            for req in dist.batch_isend_irecv(ops):
                req.wait()
    return recv, recv_counts


def classify_partitioned(stages, bounds, dist):
    """One batch on one rank of a range-partitioned index.  `stages` provides extract_sorted(bounds) ->
    (metamers [n,2] sorted, count per range[, start per range: neighbouring runs may overlap]), join(run) -> matches [m,3], score(matches) -> results;
    `bounds[p]` is the lower amino-acid-part bound of rank p's range.  Returns what score() returns for this
    rank's reads.  Communication: 2 x (count all-gather + point-to-point payload exchange); with backend "gloo" (CPU tests)
    the payload is staged through host memory."""
    import os
    import time
    import torch
    world = dist.get_world_size()
    assert len(bounds) == world, "one value range per rank"
    timing = os.environ.get("MTB_PART_TIMING")
    t = [time.perf_counter()]

    def mark():
        if timing:
            torch.cuda.synchronize() if torch.cuda.is_available() else None
            t.append(time.perf_counter())
    ext = stages.extract_sorted(bounds)
    mark()
    kmers, counts, starts = ext if len(ext) == 3 else (ext[0], ext[1], None)
    xdev = kmers.device if dist.get_backend() == "nccl" else torch.device("cpu")
    home = kmers.device
    recv, recv_counts = _exchange(torch, dist, kmers, counts, 2, xdev, starts)   # all-to-all #1: metamers to range owners
    recv = recv.to(home)
    mark()
    runs, o = [], 0
    for n in recv_counts:                                                         # runs arrive sorted: one join per source, no merge
        runs.append(stages.join(recv[o:o + n])); o += n
    m_counts = [int(r.shape[0]) for r in runs]
    m_send = runs[0] if len(runs) == 1 else (torch.cat(runs) if runs else torch.empty((0, 3), dtype=torch.int64, device=home))
    mark()
    m_recv, _ = _exchange(torch, dist, m_send, m_counts, 3, xdev)                 # all-to-all #2: matches back to the read's home
    m_recv = m_recv.to(home)
    mark()
    out = stages.score(m_recv)
    mark()
    if timing and dist.get_rank() == 0:
        import sys
        names = ("extract+sort", "exchange 1", "join", "exchange 2", "place+score+results")
        print("[partitioned] " + ", ".join(f"{n} {1e3 * (b - a):.1f} ms" for n, a, b in zip(names, t[:-1], t[1:])) +
              f"; {int(kmers.shape[0])} metamers out, {int(m_recv.shape[0])} matches home", file=sys.stderr, flush=True)
    return out
