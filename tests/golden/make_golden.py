#!/usr/bin/env python3
"""Generates tests/golden/pipeline_*.json: small seeded read sets (packed words + filtered flags)
with the expected per-read (cluster_id, keep), summary and histograms.

Provenance: the expected values come from oracle/humid_oracle.c (this repo's CPU restatement),
cross-checked at generation time against tests/bruteforce.py.  They are NOT outputs of the
reference binary: jfjlaros/HUMID cannot be built in this image (empty submodules),
and its own tests hold no end-to-end vectors (tests/Makefile:8 FIXTURES is empty).  They pin the
oracle and the HIP path against silent drift.

Run from the repo root:  python tests/golden/make_golden.py
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(HERE))

import bruteforce as bf  # noqa: E402
from humid_amd.synth import synth_words  # noqa: E402
from oracle import pyoracle as orc  # noqa: E402

CASES = [
    # name, reads, word_nt, distance, method, mode, p_sub, p_n
    ("se_umi8_d1", 600, 24, 1, 0, "umi", 2e-2, 3e-3),
    ("pe_noumi_d2", 500, 24, 2, 0, "genome", 2e-2, 3e-3),
    ("short_words_d1_max", 400, 6, 1, 1, "umi", 5e-2, 1e-2),
    ("short_words_d2", 400, 7, 2, 0, "umi", 5e-2, 1e-2),
    ("n32_d1", 300, 32, 1, 0, "umi", 2e-2, 3e-3),
    # wide words (two uint64 per read, [hi, lo]) and Levenshtein neighbours (-e)
    ("wide48_d1", 400, 48, 1, 0, "wide", 1e-2, 3e-3),
    ("wide64_d2_max", 300, 64, 2, 1, "wide", 1e-2, 3e-3),
    ("edit_d2", 400, 14, 2, 0, "indel", 0, 0),
    ("edit_d3_max", 300, 10, 3, 1, "indel", 0, 0),
]


def main():
    from humid_amd.synth import synth_wide_words
    from test_oracle_vs_bruteforce import indel_words
    for name, n_reads, n, d, method, mode, p_sub, p_n in CASES:
        edit = mode == "indel"
        if mode == "wide":
            words, filt = synth_wide_words(n_reads, 2024, n, p_sub=p_sub, p_n=p_n)
        elif edit:
            rng = np.random.default_rng(2024)
            words = indel_words(rng, n_reads, n)
            filt = (rng.random(n_reads) < 0.02).astype(np.uint8)
        else:
            words, filt = synth_words(n_reads, 2024, n, p_sub=p_sub, p_n=p_n, mode=mode, genome_bp=2000)
        p = orc.Pipeline(n)
        p.read_data(words, filt)
        if edit:
            p.find_edit_neighbours(d)
        else:
            p.find_hamming_neighbours(d)
        p.find_clusters(bool(method))
        cid, keep = p.map_reads()
        bcid, bkeep, _ = bf.dedup(words, filt, d, bool(method), edit_nt=n if edit else 0)
        assert np.array_equal(cid, bcid) and np.array_equal(keep, bkeep), name
        h = orc.histograms(p)
        doc = dict(name=name, word_nt=n, distance=d, method=method, edit=int(edit),
                   words=np.asarray(words).tolist(), filtered=[int(x) for x in filt],
                   cluster_id=[int(x) for x in cid], keep=[int(x) for x in keep],
                   summary=p.summary(), histograms={k: v for k, v in h.items()})
        with open(os.path.join(HERE, "pipeline_%s.json" % name), "w") as fh:
            json.dump(doc, fh)
        print(name, p.summary())


if __name__ == "__main__":
    main()
