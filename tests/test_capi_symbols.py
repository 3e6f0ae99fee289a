"""The C-ABI library loads on a CPU-only box and exports every function that
include/walt_amd.h declares (no compute calls here)."""
import ctypes
import os
import re

import refio


def declared_functions():
    hdr = open(os.path.join(refio.ROOT, "include", "walt_amd.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    names = re.findall(r"\b(walt_[a-z0-9_]+)\s*\(", hdr)
    return sorted(set(names))


import pytest


@pytest.mark.parametrize("pattern", [3, 5, 7])
def test_library_exports_every_declared_symbol(pattern):
    import walt_amd
    L = ctypes.CDLL(walt_amd.lib_path(pattern))
    names = declared_functions()
    assert len(names) >= 18
    for nm in names:
        assert hasattr(L, nm), "%s does not export %s" % (os.path.basename(walt_amd.lib_path(pattern)), nm)
    assert L.walt_seed_pattern() == pattern
    L.walt_min_read_len.restype = ctypes.c_uint32
    L.walt_max_read_len.restype = ctypes.c_uint32
    assert (L.walt_min_read_len(), L.walt_max_read_len()) == {3: (38, 1024), 5: (32, 148), 7: (23, 152)}[pattern]


def test_struct_layouts_match_reference_types():
    import walt_amd
    # BestMatch: 16 bytes, strand at offset 8, mismatch at 12 (mapping.hpp:39-52)
    bm = walt_amd.best_match_dtype
    assert bm.itemsize == 16 and bm.fields["strand"][1] == 8 and bm.fields["mismatch"][1] == 12
    # CandidatePosition: 12 bytes, strand at 4, mismatch at 8 (paired.hpp:35-46)
    cd = walt_amd.candidate_dtype
    assert cd.itemsize == 12 and cd.fields["strand"][1] == 4 and cd.fields["mismatch"][1] == 8


def test_no_device_is_a_loud_error_not_a_fallback(g1_index_path):
    import pytest
    import walt_amd
    if walt_amd.device_count() > 0:
        pytest.skip("GPU present")
    with pytest.raises(walt_amd.WaltError) as ei:
        walt_amd.Index.open(g1_index_path)
    assert ei.value.code == -3 and "no CPU fallback" in str(ei.value)


def test_comm_init_fails_locally_before_the_collective():
    """walt_comm_init does everything that can fail on THIS rank (device, stream, buffer) before it enters
    ncclCommInitRank (csrc/comm.hip): on a box without a GPU a rank of a two-rank world returns an error at once -- it does
    not sit in the collective waiting for a peer that will never come.  (With a GPU the call would block for the peer: skipped.)
    In a child process: the call loads librccl, and a process that later imports torch would then hold two copies of it."""
    import subprocess
    import sys

    import walt_amd
    if walt_amd.device_count() > 0:
        pytest.skip("GPU present")
    code = ("import ctypes, sys, time\n"
            "sys.path.insert(0, %r)\n"
            "import walt_amd\n"
            "L = walt_amd.lib()\n"
            "ident = ctypes.create_string_buffer(128)\n"
            "comm = ctypes.c_void_p()\n"
            "t0 = time.time()\n"
            "rc = L.walt_comm_init(0, 0, 2, ident, ctypes.byref(comm))\n"
            "assert rc != 0 and not comm.value, rc\n"
            "assert time.time() - t0 < 20\n"
            "print('ok')\n") % refio.ROOT
    pr = subprocess.run([sys.executable, "-c", code], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=120)
    assert pr.returncode == 0 and "ok" in pr.stdout, pr.stdout[-400:]
