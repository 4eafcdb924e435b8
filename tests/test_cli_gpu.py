"""-m gpu: the `humid` CLI end to end (FastQ in -> _dedup/_annotated FastQ + .dat out) against
outputs constructed from the oracle."""
import gzip
import os
import subprocess

import numpy as np
import pytest

from cli_util import HUMID, expected_words, read_fastq
from humid_amd.synth import synth_fastq
from oracle import pyoracle as orc

pytestmark = pytest.mark.gpu


def run_oracle(words, filt, n, d, maximum):
    p = orc.Pipeline(n)
    p.read_data(words, filt)
    p.find_hamming_neighbours(d)
    p.find_clusters(maximum)
    cid, keep = p.map_reads()
    return p, cid, keep


def dat(path):
    return [tuple(int(x) for x in l.split()) for l in open(path).read().strip().split("\n") if l]


@pytest.mark.parametrize("case", [
    dict(n_files=1, umi_len=8, umi_in_header=True, word_nt=24, d=1, x=False, n=3000),
    dict(n_files=2, umi_len=8, umi_in_header=True, word_nt=24, d=1, x=False, n=3000),
    dict(n_files=2, umi_len=12, umi_in_header=False, umi_file=True, word_nt=24, d=1, x=True, n=2000),
    dict(n_files=2, umi_len=0, umi_in_header=False, word_nt=12, d=2, x=False, n=2000),
    dict(n_files=2, umi_len=8, umi_in_header=True, word_nt=48, d=1, x=False, n=3000),   # wide words
    dict(n_files=2, umi_len=12, umi_in_header=False, umi_file=True, word_nt=64, d=2, x=True, n=2000),
])
def test_cli_end_to_end(case, tmp_path):
    case = dict(case)
    word_nt, d, x, n = case.pop("word_nt"), case.pop("d"), case.pop("x"), case.pop("n")
    files = synth_fastq(str(tmp_path / "in"), n, 123, p_sub=4e-3, p_n=2e-3, read_len=36,
                        short_frac=0.01, **case)
    out = str(tmp_path / "out" / "nested")
    cmd = [HUMID, "-n", str(word_nt), "-m", str(d), "-d", out, "-l", str(tmp_path / "log.txt"), "-s", "-a"]
    if x:
        cmd.append("-x")
    subprocess.check_call(cmd + files)
    words, filt, recs, _ = expected_words(files, word_nt)
    p, cid, keep = run_oracle(words, filt, word_nt, d, x)
    for fi, f in enumerate(files):
        base = os.path.basename(f)
        dedup = read_fastq(os.path.join(out, base.replace(".fastq", "_dedup.fastq")))
        annot = read_fastq(os.path.join(out, base.replace(".fastq", "_annotated.fastq")))
        assert dedup == [recs[fi][i] for i in range(n) if keep[i]]
        exp_annot = [(recs[fi][i][0] + ":%d" % cid[i],) + recs[fi][i][1:] for i in range(n)]
        assert annot == exp_annot
    h = orc.histograms(p)
    assert dat(os.path.join(out, "counts.dat")) == h["counts"]
    assert dat(os.path.join(out, "neigh.dat")) == h["neigh"]
    assert dat(os.path.join(out, "clusters.dat")) == h["clusters"]
    stats = dict(l.split(": ") for l in open(os.path.join(out, "stats.dat")).read().strip().split("\n"))
    assert {k: int(v) for k, v in stats.items()} == h["stats"]
    log = open(tmp_path / "log.txt").read()
    assert "Calculating neighbours using Hamming distance... done." in log
    assert ("Calculating maximum clusters" if x else "Calculating directional clusters") in log
    assert "Writing filtered results... done." in log and "Writing annotated results... done." in log


def test_cli_gz_roundtrip_and_q_flag(tmp_path):
    files = synth_fastq(str(tmp_path), 500, 9, n_files=1, read_len=30)
    gz = str(tmp_path / "reads.fastq.gz")
    with gzip.open(gz, "wb") as fh:
        fh.write(open(files[0], "rb").read())
    out = str(tmp_path / "o")
    subprocess.check_call([HUMID, "-d", out, "-l", "/dev/null", gz])
    d1 = read_fastq(os.path.join(out, "reads_dedup.fastq.gz"))
    words, filt, recs, _ = expected_words([gz], 24)
    _, cid, keep = run_oracle(words, filt, 24, 1, False)
    assert d1 == [recs[0][i] for i in range(len(words)) if keep[i]]
    # -q flips the default: no dedup output
    out2 = str(tmp_path / "o2")
    subprocess.check_call([HUMID, "-q", "-d", out2, "-l", "/dev/null", gz])
    assert not os.path.exists(os.path.join(out2, "reads_dedup.fastq.gz"))


def test_fast_and_streaming_host_paths_write_identical_files(tmp_path):
    files = synth_fastq(str(tmp_path / "in"), 30000, 321, n_files=2, umi_len=8, read_len=40, p_sub=4e-3,
                        p_n=2e-3)
    outs = []
    # fast: records from the mapping, words packed on the GPU, plain outputs through a shared mapping;
    # devicepack: raw symbols uploaded, words packed on the GPU; nopinned: pageable staging; bufwrite: outputs through the buffered writer; slow: streaming
    for name, env in (("fast", {"HUMID_THREADS": "5"}), ("slow", {"HUMID_HOST_SLOW": "1"}),
                      ("devicepack", {"HUMID_DEVICE_PACK": "1", "HUMID_THREADS": "3"}), ("nopinned", {"HUMID_NO_PINNED": "1"}),
                      ("bufwrite", {"HUMID_NO_MAPPED_WRITE": "1"}), ("onethread", {"HUMID_THREADS": "1"})):
        out = str(tmp_path / name)
        e = dict(os.environ)
        e.update(env)
        subprocess.check_call([HUMID, "-d", out, "-l", "/dev/null", "-a", "-s"] + files, env=e)
        outs.append(out)
    for other in outs[1:]:
        for fn in sorted(os.listdir(outs[0])):
            assert open(os.path.join(outs[0], fn), "rb").read() == open(os.path.join(other, fn), "rb").read(), (other, fn)
    assert len(os.listdir(outs[0])) == 8     # 2 dedup + 2 annotated + 4 .dat


def test_cli_gzip_fast_path_outputs(tmp_path):
    """gzip in -> gzip out through the inflate-once / parallel-member path: decompressed outputs
    equal the streaming path's and the oracle-derived expectation, for dedup and annotated files"""
    files = synth_fastq(str(tmp_path / "in"), 6000, 31, n_files=2, umi_len=8, read_len=36, p_sub=4e-3, p_n=2e-3)
    gzs = []
    for f in files:
        g = f + ".gz"
        with gzip.open(g, "wb") as fh:
            fh.write(open(f, "rb").read())
        gzs.append(g)
    outs = {}
    for label, env in (("fast", {"HUMID_THREADS": "6"}), ("slow", {"HUMID_HOST_SLOW": "1"})):
        out = str(tmp_path / ("out_" + label))
        e = dict(os.environ)
        e.update(env)
        subprocess.check_call([HUMID, "-d", out, "-l", "/dev/null", "-a"] + gzs, env=e)
        outs[label] = out
    words, filt, recs, _ = expected_words(files, 24)
    _, cid, keep = run_oracle(words, filt, 24, 1, False)
    for fi, g in enumerate(gzs):
        base = os.path.basename(g)
        for suffix in ("_dedup", "_annotated"):
            name = base.replace(".fastq.gz", suffix + ".fastq.gz")
            a = gzip.open(os.path.join(outs["fast"], name), "rb").read()
            b = gzip.open(os.path.join(outs["slow"], name), "rb").read()
            assert a == b, name
        dedup = read_fastq(os.path.join(outs["fast"], base.replace(".fastq.gz", "_dedup.fastq.gz")))
        assert dedup == [recs[fi][i] for i in range(len(words)) if keep[i]]
    # nothing kept (every read filtered): the member-mode writer still leaves a valid, empty gzip
    nf = str(tmp_path / "n.fastq.gz")
    with gzip.open(nf, "wb") as fh:
        for i in range(50):
            fh.write(b"@r%d_NNNNNNNN\nNNNNNNNNNNNNNNNNNNNN\n+\nIIIIIIIIIIIIIIIIIIII\n" % i)
    out = str(tmp_path / "out_n")
    subprocess.check_call([HUMID, "-d", out, "-l", "/dev/null", nf])
    assert gzip.open(os.path.join(out, "n_dedup.fastq.gz"), "rb").read() == b""


def test_cli_edit_distance(tmp_path):
    """-e -m 2: Levenshtein neighbours end to end; the log names the distance (src/humid.cc:142)"""
    files = synth_fastq(str(tmp_path / "in"), 3000, 77, n_files=1, umi_len=8, read_len=36, p_sub=2e-2)
    out = str(tmp_path / "out")
    subprocess.check_call([HUMID, "-e", "-m", "2", "-d", out, "-l", str(tmp_path / "log.txt"), "-a", "-s"] + files)
    words, filt, recs, _ = expected_words(files, 24)
    p = orc.Pipeline(24)
    p.read_data(words, filt)
    p.find_edit_neighbours(2)
    p.find_clusters(False)
    cid, keep = p.map_reads()
    base = os.path.basename(files[0])
    dedup = read_fastq(os.path.join(out, base.replace(".fastq", "_dedup.fastq")))
    annot = read_fastq(os.path.join(out, base.replace(".fastq", "_annotated.fastq")))
    assert dedup == [recs[0][i] for i in range(len(words)) if keep[i]]
    assert annot == [(recs[0][i][0] + ":%d" % cid[i],) + recs[0][i][1:] for i in range(len(words))]
    assert dat(os.path.join(out, "neigh.dat")) == orc.histograms(p)["neigh"]
    assert "Calculating neighbours using Levenshtein distance... done." in open(tmp_path / "log.txt").read()


def test_cli_output_write_errors_are_not_silent(tmp_path):
    """a full disk (here: outputs that resolve to /dev/full) must end in a non-zero exit code, for
    plain and for gzip outputs (short writes were swallowed, exit code 0)"""
    if not os.path.exists("/dev/full"):
        pytest.skip("no /dev/full")
    files = synth_fastq(str(tmp_path / "in"), 3000, 5, n_files=1, read_len=36)
    for gz in (False, True):
        src = files[0]
        if gz:
            src = files[0] + ".gz"
            with gzip.open(src, "wb") as fh:
                fh.write(open(files[0], "rb").read())
        out = tmp_path / ("out_gz" if gz else "out")
        out.mkdir()
        name = os.path.basename(src).replace(".fastq", "_dedup.fastq")
        os.symlink("/dev/full", str(out / name))
        p = subprocess.run([HUMID, "-d", str(out), "-l", str(tmp_path / "log.txt"), src],
                           stderr=subprocess.PIPE)
        assert p.returncode == 1, p.stderr
        assert b"failed" in p.stderr
        # the same command with a writable output succeeds
        os.unlink(str(out / name))
        assert subprocess.call([HUMID, "-d", str(out), "-l", str(tmp_path / "log.txt"), src]) == 0


@pytest.mark.parametrize("case", [
    dict(n_files=2, umi_len=8, umi_in_header=True, word_nt=24, d=1, x=False, n=40_000),
    dict(n_files=2, umi_len=8, umi_in_header=True, word_nt=48, d=2, x=True, n=20_000),
])
def test_cli_sixteen_ranks(case, tmp_path):
    """`humid -g 16`, the largest group the exchange pass takes (the library lost a counter at exactly 16 ranks until the
    end of round 3: tests/test_gpu_exchange.py::test_exchange_sixteen_ranks): every output byte-identical to `-g 1`"""
    case = dict(case)
    word_nt, d, x, n = case.pop("word_nt"), case.pop("d"), case.pop("x"), case.pop("n")
    files = synth_fastq(str(tmp_path / "in"), n, 78, p_sub=4e-3, p_n=2e-3, read_len=36, short_frac=0.01, **case)
    outs = {}
    for g in (1, 16):
        out = str(tmp_path / ("out%d" % g))
        cmd = [HUMID, "-n", str(word_nt), "-m", str(d), "-d", out, "-l", "/dev/null", "-s", "-a", "-g", str(g)] + (["-x"] if x else [])
        r = subprocess.run(cmd + files, capture_output=True, text=True, env=dict(os.environ, HUMID_TIMING="1"))
        assert r.returncode == 0, r.stderr
        if g > 1:
            assert "16 ranks, bulk data by copy" in r.stderr
        outs[g] = {f: open(os.path.join(out, f), "rb").read() for f in sorted(os.listdir(out))}
    assert sorted(outs[1]) == sorted(outs[16]) and len(outs[1]) == 2 * len(files) + 4
    for f in outs[1]:
        assert outs[1][f] == outs[16][f], f


@pytest.mark.parametrize("ranks", [2, 3])
@pytest.mark.parametrize("case", [
    dict(n_files=2, umi_len=8, umi_in_header=True, word_nt=24, d=1, x=False, n=40_000),
    dict(n_files=1, umi_len=8, umi_in_header=True, word_nt=24, d=2, x=True, n=12_000),
    dict(n_files=2, umi_len=0, umi_in_header=False, word_nt=12, d=1, x=False, n=9_000),
    dict(n_files=2, umi_len=8, umi_in_header=True, word_nt=48, d=1, x=False, n=20_000),    # two-word (wide) words
    dict(n_files=2, umi_len=12, umi_in_header=False, umi_file=True, word_nt=64, d=2, x=True, n=6_000),
])
def test_cli_sharded_over_ranks_writes_the_single_gpu_files(case, ranks, tmp_path):
    """`humid -g N` (csrc/host/sharded.cpp: the exchange orchestration driven from the C++ host, one
    rank per shard of the reads in input order).  The test box has ONE GPU, so the ranks share it and
    exchange through peer copies (HUMID_COMM=copy is what -g picks by itself then); every output file --
    deduplicated and annotated FastQ, the three histograms, stats.dat -- must be byte-identical to the
    single-GPU run's, and the single-GPU run is checked against the oracle."""
    case = dict(case)
    word_nt, d, x, n = case.pop("word_nt"), case.pop("d"), case.pop("x"), case.pop("n")
    files = synth_fastq(str(tmp_path / "in"), n, 77, p_sub=4e-3, p_n=2e-3, read_len=36, short_frac=0.01, **case)
    outs = {}
    for g in (1, ranks):
        out = str(tmp_path / ("out%d" % g))
        cmd = [HUMID, "-n", str(word_nt), "-m", str(d), "-d", out, "-l", str(tmp_path / ("log%d.txt" % g)),
               "-s", "-a", "-g", str(g)]
        if x:
            cmd.append("-x")
        r = subprocess.run(cmd + files, capture_output=True, text=True, env=dict(os.environ, HUMID_TIMING="1"))
        assert r.returncode == 0, r.stderr
        if g > 1:
            assert "%d ranks, bulk data by copy" % g in r.stderr
        outs[g] = {f: open(os.path.join(out, f), "rb").read() for f in sorted(os.listdir(out))}
    assert sorted(outs[1]) == sorted(outs[ranks]) and len(outs[1]) == 2 * len(files) + 4
    for f in outs[1]:
        assert outs[1][f] == outs[ranks][f], f
    words, filt, recs, _ = expected_words(files, word_nt)
    p, cid, keep = run_oracle(words, filt, word_nt, d, x)
    base = os.path.basename(files[0])
    annot = read_fastq(os.path.join(str(tmp_path / ("out%d" % ranks)), base.replace(".fastq", "_annotated.fastq")))
    assert annot == [(recs[0][i][0] + ":%d" % cid[i],) + recs[0][i][1:] for i in range(n)]


def test_cli_sharded_rccl_transport_with_one_rank(tmp_path):
    """the RCCL transport of `-g` (grouped ncclSend/ncclRecv on the context's stream, librccl loaded on
    demand) needs a GPU per rank.  The test box has one, so the transport runs here with ONE rank
    (HUMID_FORCE_SHARDED=1: the rank orchestration instead of the single-GPU call; every exchange is a
    send to and a receive from rank 0 itself) and must write the single-GPU files; two ranks on one GPU
    are refused.  More ranks over RCCL run on the multi-GPU node only."""
    files = synth_fastq(str(tmp_path / "in"), 20_000, 5, n_files=2, umi_len=8, umi_in_header=True, read_len=36,
                        p_sub=4e-3, p_n=2e-3)
    outs = {}
    for tag, env in (("plain", {}), ("rccl", dict(HUMID_FORCE_SHARDED="1", HUMID_COMM="rccl", HUMID_TIMING="1"))):
        out = str(tmp_path / tag)
        r = subprocess.run([HUMID, "-d", out, "-l", "/dev/null", "-s", "-a"] + files, capture_output=True, text=True,
                           env=dict(os.environ, **env))
        assert r.returncode == 0, r.stderr
        if tag == "rccl":
            assert "1 ranks, bulk data by rccl" in r.stderr, r.stderr
        outs[tag] = {f: open(os.path.join(out, f), "rb").read() for f in sorted(os.listdir(out))}
    assert outs["plain"] == outs["rccl"]
    import torch
    if torch.cuda.device_count() < 2:
        r = subprocess.run([HUMID, "-g", "2", "-d", str(tmp_path / "o"), "-l", "/dev/null"] + files, capture_output=True,
                           text=True, env=dict(os.environ, HUMID_COMM="rccl"))
        assert r.returncode == 1 and "one GPU per rank" in r.stderr


@pytest.mark.parametrize("case", [dict(word_nt=24, d=2, x=False, ranks=2), dict(word_nt=24, d=3, x=True, ranks=3),
                                  dict(word_nt=40, d=2, x=False, ranks=2), dict(word_nt=20, d=4, x=False, ranks=2),
                                  dict(word_nt=24, d=6, x=False, ranks=2), dict(word_nt=24, d=7, x=True, ranks=3)])
def test_cli_sharded_edit_distance(case, tmp_path):
    """round 3: `humid -g N -e -m 2...` (beyond 5 since the end of the round) -- the unique words are all-gathered inside the
    exchange pass and the shifted-segment joins dealt out over the ranks.  Every output file byte-identical to
    `-g 1 -e` (which tests/test_gpu_edit.py and test_cli_end_to_end check against the oracle)."""
    files = synth_fastq(str(tmp_path / "in"), 4000, 91, n_files=2, umi_len=8, umi_in_header=True, read_len=36,
                        p_sub=8e-3, p_n=2e-3, short_frac=0.01)
    outs = {}
    for g in (1, case["ranks"]):
        out = str(tmp_path / ("out%d" % g))
        cmd = [HUMID, "-n", str(case["word_nt"]), "-m", str(case["d"]), "-e", "-d", out, "-l", "/dev/null", "-s", "-a", "-g", str(g)]
        if case["x"]:
            cmd.append("-x")
        r = subprocess.run(cmd + files, capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        outs[g] = {f: open(os.path.join(out, f), "rb").read() for f in sorted(os.listdir(out))}
    assert sorted(outs[1]) == sorted(outs[case["ranks"]])
    for f in outs[1]:
        assert outs[1][f] == outs[case["ranks"]][f], f


def test_cli_sharded_refuses_what_needs_one_gpu(tmp_path):
    files = synth_fastq(str(tmp_path / "in"), 200, 5, n_files=1, read_len=30)
    for extra in (["-g", "17"],):
        r = subprocess.run([HUMID, "-g", "2", "-d", str(tmp_path / "o"), "-l", "/dev/null"] + extra + files,
                           capture_output=True, text=True)
        assert r.returncode == 2, (extra, r.stderr)


@pytest.mark.parametrize("n_reads", [0, 1, 2, 7])
def test_cli_sharded_tiny_inputs(n_reads, tmp_path):
    """fewer reads than ranks, empty shards, an empty file: `-g 3` still writes what `-g 1` writes"""
    if n_reads:
        files = synth_fastq(str(tmp_path / "in"), n_reads, 3, n_files=1, umi_len=8, umi_in_header=True, read_len=30)
    else:
        os.makedirs(tmp_path / "in")
        files = [str(tmp_path / "in" / "empty.fastq")]
        open(files[0], "w").close()
    outs = {}
    for g in (1, 3):
        out = str(tmp_path / ("o%d" % g))
        r = subprocess.run([HUMID, "-g", str(g), "-d", out, "-l", "/dev/null", "-s", "-a"] + files, capture_output=True, text=True)
        assert r.returncode == 0, (g, r.stderr)
        outs[g] = {f: open(os.path.join(out, f), "rb").read() for f in sorted(os.listdir(out))}
    assert outs[1] == outs[3]
