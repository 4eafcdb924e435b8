"""The oracle against every known answer the reference's own tests hold
(tests/golden/ref_test_cluster.json <- /root/reference/tests/test_cluster.cc,
 tests/golden/ref_test_fastq.json   <- /root/reference/tests/test_fastq.cc)."""
import json
import os

import numpy as np
import pytest

from oracle import pyoracle as orc


@pytest.fixture(scope="module")
def gc(golden_dir):
    return json.load(open(os.path.join(golden_dir, "ref_test_cluster.json")))


@pytest.fixture(scope="module")
def gf(golden_dir):
    return json.load(open(os.path.join(golden_dir, "ref_test_fastq.json")))


def test_at_least_double(gc):
    for c in gc["at_least_double"]:
        assert bool(orc.lib().orc_at_least_double(c["a"], c["b"])) == c["expect"], c["ref"]


def test_max_neighbour(gc):
    for case in gc["max_neighbour"]:
        g = orc.Graph(case["counts"])
        for a, b in case["links"]:
            g.link(a, b)
        for leaf, cid in case["preassigned"].items():
            g.preassign(int(leaf), cid)
        for q in case["queries"]:
            assert g.max_neighbour(q["leaf"]) == q["expect"], case["name"]


def test_assign_directional(gc):
    case = gc["assign_directional"]
    g = orc.Graph(case["counts"])
    for a, b in case["links"]:
        g.link(a, b)
    for call in case["calls"]:
        g.assign(call["leaf"], call["cluster"], maximum=False)
        lc, _, _, _ = g.export(2)
        assert lc.tolist() == call["expect_leaf_cluster"]
    lc, size, mc, ml = g.export(2)
    assert size.tolist() == case["expect_size"]
    assert ml.tolist() == case["expect_max_leaf"]
    assert mc.tolist() == case["expect_max_count"]


def test_assign_directional_via_find_clusters_loop(gc):
    # the same scenario driven by the findClusters loop (src/humid.cc:176-189):
    # leaf 0 creates cluster 1, leaf 3 is the next unassigned leaf -> cluster 2
    case = gc["assign_directional"]
    g = orc.Graph(case["counts"])
    for a, b in case["links"]:
        g.link(a, b)
    assert g.find_clusters(False) == 2
    lc, size, mc, ml = g.export(2)
    assert lc.tolist() == case["calls"][-1]["expect_leaf_cluster"]
    assert size.tolist() == case["expect_size"]


def test_extract_umi(gf):
    for c in gf["extract_umi"]:
        assert orc.extract_umi(c["header"]) == c["expect"], c["ref"]


def test_make_word(gf):
    for c in gf["make_word"]:
        nuc = orc.get_nucleotides(c["headers"][0], c["seqs"], c["nt_to_take"], c["header_umi_size"])
        data, filt = orc.make_word(nuc)
        assert data == c["expect_data"] and filt == c["expect_filtered"], c["ref"]


def test_get_nucleotides(gf):
    for c in gf["get_nucleotides"]:
        got = orc.get_nucleotides(c["headers"][0], c["seqs"], c["nt_to_take"], c["header_umi_size"])
        assert got == c["expect"], c["ref"]


def test_nt_from_file(gf):
    for c in gf["nt_from_file"]:
        assert orc.nt_from_file(c["files"], c["length"]) == c["expect"], c["ref"]


def test_valid_umi(gf):
    for c in gf["valid_umi"]:
        assert orc.valid_umi(c["umi"]) == c["expect"], c["ref"]


def test_extract_last_field(gf):
    for c in gf["extract_last_field"]:
        assert orc.extract_last_field(c["str"], c["sep"]) == c["expect"], c["ref"]


def test_make_string_size(gf):
    for c in gf["make_string_size"]:
        assert orc.make_string_size(c["s"], c["size"], c["pad"]) == c["expect"], c["ref"]


def test_make_word_unknown_base_is_G_and_filtered():
    # src/fastq.cc:151-158: anything outside ACGT pushes the code of 'G' and filters
    data, filt = orc.make_word("ACGTNacgt")
    assert data == [0, 1, 2, 3, 2, 2, 2, 2, 2] and filt


def test_pack_word_is_lexicographic():
    assert orc.pack_word([0, 0, 0, 0, 3, 3, 3, 3]) == 0x00FF
    assert orc.pack_word([3, 0]) > orc.pack_word([2, 3])


def test_pre_compute_matches_usage_doc():
    # docs/usage.rst:6-11: three files, -n 23 -> 7, 7, 9
    assert orc.pre_compute(0, 3, 23) == (0, [7, 7, 9])
    # header UMI longer than the word is clamped (src/humid.cc:54-56)
    assert orc.pre_compute(30, 2, 24) == (24, [0, 0])
    assert orc.pre_compute(8, 1, 24) == (8, [16])
    assert orc.pre_compute(8, 2, 24) == (8, [8, 8])
