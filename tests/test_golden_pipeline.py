"""Committed pipeline goldens (tests/golden/pipeline_*.json, made by tests/golden/make_golden.py):
the oracle on CPU, the HIP path on the GPU."""
import glob
import json
import os

import numpy as np
import pytest

from oracle import pyoracle as orc

GOLD = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "pipeline_*.json")))


def load(path):
    g = json.load(open(path))
    g["words"] = np.array(g["words"], dtype=np.uint64)
    g["filtered"] = np.array(g["filtered"], dtype=np.uint8)
    g["histograms"] = {k: ([tuple(x) for x in v] if isinstance(v, list) else v)
                       for k, v in g["histograms"].items()}
    return g


def test_goldens_exist():
    assert len(GOLD) >= 9


@pytest.mark.parametrize("path", GOLD, ids=[os.path.basename(p) for p in GOLD])
def test_oracle_matches_golden(path):
    g = load(path)
    cid, keep, s, _ = orc.dedup_run(g["words"], g["filtered"], g["word_nt"], g["distance"], g["method"],
                                    edit=bool(g.get("edit", 0)))
    assert cid.tolist() == g["cluster_id"] and keep.tolist() == g["keep"]
    for k in ("total", "usable", "unique", "clusters"):
        assert s[k] == g["summary"][k]


@pytest.mark.gpu
@pytest.mark.parametrize("path", GOLD, ids=[os.path.basename(p) for p in GOLD])
def test_hip_matches_golden(path):
    import humid_amd
    g = load(path)
    dd = humid_amd.Dedup()
    cid, keep, s = dd.run(g["words"], g["filtered"], g["word_nt"], g["distance"], g["method"],
                          edit=bool(g.get("edit", 0)))
    assert cid.tolist() == g["cluster_id"] and keep.tolist() == g["keep"]
    for k in ("total", "usable", "unique", "clusters", "edges"):
        assert s[k] == g["summary"][k]
    h = dd.histograms()
    assert h["counts"] == g["histograms"]["counts"]
    assert h["neigh"] == g["histograms"]["neigh"]
    assert h["clusters"] == g["histograms"]["clusters"]
    dd.close()
