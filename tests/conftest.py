import os
import shutil
import sys
import tempfile

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (ROOT, HERE):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _scratch_dir():
    base = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else None
    return tempfile.mkdtemp(prefix="walt_amd_test_", dir=base)


@pytest.fixture(scope="session")
def scratch():
    d = _scratch_dir()
    yield d
    shutil.rmtree(d, ignore_errors=True)


@pytest.fixture(scope="session")
def g1_index_path(scratch):
    """g1.fa indexed by the product's own makedb-compatible host builder
    (walt_amd/csrc/host_index.cpp, compiled into the test harness so that no
    GPU library is needed on the CPU box)."""
    import refio
    out = os.path.join(scratch, "g1.dbindex")
    rc = refio.harness().walt_makedb(os.path.join(refio.GOLDEN, "g1.fa").encode(), out.encode(), 4)
    assert rc == 0, refio.harness().walt_last_error()
    return out


@pytest.fixture(scope="session")
def g1_db(g1_index_path):
    import refio
    return refio.DbIndex(g1_index_path)


@pytest.fixture
def index_options():
    """index_options(idx, name=value, ...): sets options of a (shared) index for this test and puts the old values back
    afterwards (walt_index_set_option: tuning values and test hooks; they change a call's schedule, never its results)."""
    undo = []

    def setter(idx, **kv):
        for name, value in kv.items():
            undo.append((idx, name, idx.get_option(name)))
            idx.set_option(name, value)

    yield setter
    for idx, name, old in reversed(undo):
        idx.set_option(name, old)


@pytest.fixture(scope="module")
def wa_diag():
    """walt_amd with its diagnostic library loaded as well (walt_amd.diag_lib(): libwalt_amd_diag.so)."""
    import walt_amd
    assert walt_amd.device_count() >= 1, "no HIP device: the walt_amd hot path has no CPU fallback"
    walt_amd.diag_lib()
    return walt_amd
