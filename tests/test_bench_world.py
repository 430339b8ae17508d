"""bench.py's synthetic world (CPU, torch on the host): the HELD-OUT genomes of the novel leg come from a generator of their own -- the indexed
world is bit-identical with and without them -- and are what the leg's description says: the first species of a genus with ~7.5 %
substitutions, the conserved segments the parent carries drawn again."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_heldout_genomes_leave_the_indexed_world_untouched():
    import torch
    import bench
    dev = torch.device("cpu")
    kw = dict(seed=77, n_species=24, genome_len=650_000, n_filler_species=50, conserved=True)
    w0 = bench.build_world_fast(torch, dev, kw["seed"], kw["n_species"], kw["genome_len"], kw["n_filler_species"], conserved=True)
    w1 = bench.build_world_fast(torch, dev, kw["seed"], kw["n_species"], kw["genome_len"], kw["n_filler_species"], conserved=True, n_heldout=4)
    assert len(w0.genomes) == len(w1.genomes) == 24 and not w0.heldout and len(w1.heldout) == 4
    for (t0, g0), (t1, g1) in zip(w0.genomes, w1.genomes):
        assert t0 == t1 and np.array_equal(g0, g1)                      # same indexed world
    assert w0.tax.parent == w1.tax.parent
    by_tid = dict(w1.genomes)
    for parent_tid, h in w1.heldout:
        g = by_tid[parent_tid]
        assert len(h) == len(g) and set(np.unique(h)) <= set(b"ACGT")
        diff = float((h != g).mean())
        # 7.5 % substitutions everywhere + the conserved segments (a sixth of the genome at this size) redrawn: synonymous codons differ in ~1 base of 3
        assert 0.07 < diff < 0.20, diff
    parents = [t for t, _ in w1.heldout]
    assert len(set(parents)) == 4 and parents == [w1.genomes[4 * i][0] for i in range(4)]      # the first species of the first four genera
