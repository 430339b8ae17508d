#!/usr/bin/env python3
"""Round-3 experiment: the slot buffer in physically contiguous VRAM (hipExtMallocWithFlags + hipDeviceMallocContiguous) made
tests/test_gpu_parity.py::test_register_resident_scorer_variants[True-110|150] disagree with the oracle in round 2 (pair scores,
not classifications).  This script replays that test's five batches on ONE context (the pytest session shares one) under a set of
environment / library variants, each in its own process, and prints which of them still disagree and how.

    python profiles/scripts/contig_diag.py            # all variants -> table on stdout
    python profiles/scripts/contig_diag.py --child    # one process, variant taken from the environment
"""
import json
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
CASES = [(False, 100), (False, 250), (True, 75), (True, 110), (True, 150)]


def child():
    import metabuli_amd as M
    from helpers import Oracle, default_params, build_toy_db
    from metabuli_amd import synth
    orc = Oracle()
    ctx = M.Context(0)
    out = []
    reps = int(os.environ.get("DIAG_REPS", "1"))
    for paired, length in CASES:
        p = default_params(seq_mode=2 if paired else 1, syncmer=1)
        world = synth.make_world(seed=40 + length, n_genera=3, species_per_genus=2, strains_per_species=1, genome_len=30000, genus_div=0.3)
        d = tempfile.mkdtemp(prefix="contig_")
        build_toy_db(orc, world, p, d)
        tax = orc.load_taxonomy(os.path.join(d, "taxonomy"))
        db = orc.open_db(d, tax, p)
        smp = synth.sample_reads(np.random.default_rng(length), world, 600, length=length, err=0.01, with_n=0.05, paired=paired, lognormal=False)
        if paired:
            b1, o1, b2, o2, _ = smp
        else:
            (b1, o1, _), b2, o2 = smp, None, None
        ref = orc.classify(db, tax, p, b1, o1, b2, o2)
        mp = M.default_params(seq_mode=p.seq_mode, syncmer=1)
        ix = ctx.open_index(d, mp)
        for rep in range(reps):
            res, tt, tc = ctx.classify_batch(ix, mp, b1, o1, b2, o2)
            ro = ref["results"]
            amb = ro["flag"] != 0
            bad_cls = (res["classification"] != ro["classification"]) & ~amb
            bad_sc = (res["score"].view(np.uint32) != ro["score"].view(np.uint32)) & ~amb
            bad_n = (res["n_taxcnt"] != ro["n_taxcnt"]) & ~amb
            st = ctx.last_stats()
            rec = dict(case=f"{'PE' if paired else 'SE'}{length}", rep=rep, bad_cls=int(bad_cls.sum()), bad_score=int(bad_sc.sum()), bad_ntc=int(bad_n.sum()),
                       matches=int(st.n_matches), oracle_matches=int(len(ref["matches"])), generic=int(st.n_generic_reads))
            if bad_sc.any():
                idx = np.flatnonzero(bad_sc)[:6]
                seq = ((ref["matches"]["qinfo"] >> np.uint64(32)) & np.uint64(0x1FFFFFFF)).astype(np.int64) - 1
                rec["examples"] = [dict(read=int(i), gpu=float(res["score"][i]), oracle=float(ro["score"][i]), oracle_matches=int((seq == i).sum()),
                                        gpu_cls=int(res["classification"][i]), orc_cls=int(ro["classification"][i])) for i in idx]
                rec["gpu_gt_oracle"] = int((res["score"][bad_sc] > ro["score"][bad_sc]).sum())
            out.append(rec)
        ix.close()
    print("DIAG " + json.dumps(out))


def main():
    if "--child" in sys.argv:
        return child()
    csrc = os.path.join(ROOT, "metabuli_amd", "csrc")
    variants = [
        ("baseline (hipMalloc)", {}),
        ("contig", {"MTB_SEGM_CONTIG": "1"}),
        ("contig + clear by kernel", {"MTB_SEGM_CONTIG": "1", "MTB_SEGM_CLEAR": "kernel"}),
        ("contig + memset + device sync", {"MTB_SEGM_CONTIG": "1", "MTB_SEGM_CLEAR": "sync"}),
        ("contig + clear every batch", {"MTB_SEGM_CONTIG": "1", "MTB_SEGM_CLEAR": "always"}),
        ("contig + 1 MB pad behind the buffer", {"MTB_SEGM_CONTIG": "1", "MTB_SEGM_PAD": "1"}),
        ("contig + pairs on the generic scorer", {"MTB_SEGM_CONTIG": "1", "MTB_NO_FAST_PAIRS": "1"}),
        ("contig + plain slot stores", {"MTB_SEGM_CONTIG": "1", "MTB_LIB": os.path.join(csrc, "libmtb_xnont.so")}),
        ("contig + poisoned buffers", {"MTB_SEGM_CONTIG": "1", "MTB_LIB": os.path.join(csrc, "libmtb_xpoison.so")}),
        ("contig, every batch twice", {"MTB_SEGM_CONTIG": "1", "DIAG_REPS": "2"}),
        ("contig + system-scope release after the slot writers / acquire before the readers", {"MTB_SEGM_CONTIG": "1", "MTB_LIB": os.path.join(csrc, "libmtb_xfence.so")}),
        ("contig + fences + clear by kernel", {"MTB_SEGM_CONTIG": "1", "MTB_SEGM_CLEAR": "kernel", "MTB_LIB": os.path.join(csrc, "libmtb_xfence.so")}),
        ("hipMalloc + fences (control)", {"MTB_LIB": os.path.join(csrc, "libmtb_xfence.so")}),
        ("contig + release fences only", {"MTB_SEGM_CONTIG": "1", "MTB_SEGM_CLEAR": "kernel", "MTB_LIB": os.path.join(csrc, "libmtb_xfencerel.so")}),
        ("contig + acquire fences only", {"MTB_SEGM_CONTIG": "1", "MTB_SEGM_CLEAR": "kernel", "MTB_LIB": os.path.join(csrc, "libmtb_xfenceacq.so")}),
    ]
    only = [a for a in sys.argv[1:] if not a.startswith("--")]
    for name, env in variants:
        if only and not any(o in name for o in only):
            continue
        if "MTB_LIB" in env and not os.path.exists(env["MTB_LIB"]):
            print(f"== {name}: {env['MTB_LIB']} not built, skipped"); continue
        p = subprocess.run([sys.executable, os.path.abspath(__file__), "--child"], env=dict(os.environ, **env), capture_output=True, text=True, timeout=900)
        line = [l for l in p.stdout.splitlines() if l.startswith("DIAG ")]
        if not line:
            print(f"== {name}: FAILED rc={p.returncode}\n{p.stderr[-800:]}"); continue
        recs = json.loads(line[0][5:])
        bad = [r for r in recs if r["bad_cls"] or r["bad_score"] or r["bad_ntc"] or r["matches"] != r["oracle_matches"]]
        print(f"== {name}: {'OK' if not bad else 'MISMATCH'}  " + " ".join(f"{r['case']}:{r['bad_score']}/{r['bad_cls']}" for r in recs))
        for r in bad:
            print("   ", json.dumps(r))
        sys.stdout.flush()


if __name__ == "__main__":
    main()
