"""Multi-GPU decomposition of the classify hot path (first row).

Reads are independent units: with the index replicated on every GPU the path
shards by contiguous read ranges and needs NO data-path collective.  The only
communication is control-plane: a barrier around timed regions and the sum of
the per-taxon read counts (Classifier.cpp:201-203) at the end.  One process
per GPU; `torch.distributed` backend "nccl" (= RCCL) on GPUs, "gloo" in the CPU
tests.
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
