"""helpers for the `humid` CLI tests: expected words / outputs derived with the ORACLE
(oracle/pyoracle.py word extraction + pipeline), never with the product code."""
import gzip
import os
import subprocess

import numpy as np

from oracle import pyoracle as orc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HUMID = os.path.join(ROOT, "humid_amd", "humid")


def read_fastq(path):
    op = gzip.open if path.endswith(".gz") else open
    with op(path, "rt") as fh:
        lines = [l.rstrip("\n") for l in fh]
    return [tuple(lines[i:i + 4]) for i in range(0, len(lines) - 3, 4)]


def expected_words(files, word_nt):
    """oracle restatement of preCompute + makeWord over whole files"""
    recs = [read_fastq(f) for f in files]
    n = min(len(r) for r in recs)
    first_umi = len(orc.extract_umi(recs[0][0][0])) if n else 0
    hdr, take = orc.pre_compute(first_umi, len(files), word_nt)
    wide = word_nt > 32      # two uint64 per word: [first word_nt-32 symbols, last 32]
    words = np.zeros((n, 2) if wide else n, dtype=np.uint64)
    filt = np.zeros(n, dtype=np.uint8)
    for i in range(n):
        nuc = orc.get_nucleotides(recs[0][i][0], [r[i][1] for r in recs], take, hdr)
        data, fl = orc.make_word(nuc)
        if wide:
            words[i, 0] = orc.pack_word(data[:word_nt - 32])
            words[i, 1] = orc.pack_word(data[word_nt - 32:])
        else:
            words[i] = orc.pack_word(data)
        filt[i] = fl
    return words, filt, recs, (hdr, take)


def dump_words(files, word_nt, tmp, env=None):
    out = os.path.join(tmp, "words.bin")
    e = dict(os.environ)
    e.update(env or {})
    subprocess.check_call([HUMID, "-n", str(word_nt), "-l", os.path.join(tmp, "log.txt"),
                           "--dump-words", out] + list(files), env=e)
    raw = open(out, "rb").read()
    n = int(np.frombuffer(raw[:8], dtype=np.uint64)[0])
    wpr = 2 if word_nt > 32 else 1
    words = np.frombuffer(raw[8:8 + 8 * n * wpr], dtype=np.uint64)
    if wpr == 2:
        words = words.reshape(n, 2)
    filt = np.frombuffer(raw[8 + 8 * n * wpr:8 + 8 * n * wpr + n], dtype=np.uint8)
    return words, filt
