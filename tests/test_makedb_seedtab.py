"""Host-side pins: seed tables vs the golden dump of the reference header, and
the makedb-compatible builder vs the md5 of the reference makedb's files."""
import hashlib
import json
import os

import numpy as np

import refio


def test_seed_tables_match_golden_dump():
    with open(os.path.join(refio.GOLDEN, "seedpattern3.json")) as f:
        gold = json.load(f)
    care = np.zeros(60, dtype=np.uint32)
    nocare = np.zeros((3, 150), dtype=np.uint32)
    refio.oracle().orc_get_tables(care.ctypes.data, nocare.ctypes.data)
    assert care.tolist() == gold["F2CAREDPOSITION"]
    assert nocare.tolist() == gold["F2NOCAREDPOSITION"]
    prod = np.zeros((3, 150), dtype=np.uint32)
    refio.harness().hh_get_nocare(prod.ctypes.data)
    assert prod.tolist() == gold["F2NOCAREDPOSITION"]


def test_seed_tables_match_reference_header_when_present():
    import pytest
    hdr = "/root/reference/src/walt/seedpattern.hpp"
    if not os.path.exists(hdr):
        pytest.skip("reference checkout not present")
    import sys
    sys.path.insert(0, refio.GOLDEN)
    import make_seedtab_golden
    with open(os.path.join(refio.GOLDEN, "seedpattern3.json")) as f:
        assert json.load(f) == make_seedtab_golden.parse(hdr)


def test_makedb_is_byte_identical_to_reference(g1_index_path):
    meta = refio.golden_meta()
    for sfx, want in meta["index_md5"].items():
        h = hashlib.md5()
        with open(g1_index_path + sfx, "rb") as f:
            for blk in iter(lambda: f.read(1 << 22), b""):
                h.update(blk)
        assert h.hexdigest() == want, sfx
