import os
import sys

import numpy as np
import pytest

# Before anything initialises HIP.  The test suite moves arrays of up to 200 MB between numpy and the device through torch's pageable copies.
# For every pageable copy of ~2 MiB and more the HIP runtime page-locks the buffer in place and caches the mapping (GPU_PINNED_MIN_XFER_SIZE;
# tools/micro/pageable_copy.py), and on this stack those cached mappings of heap memory later fault ("Memory access fault by GPU ... Write
# access to a read-only page" on a host heap address, in whatever copy comes next): 7 of 10 full `-m gpu` runs died that way, 0 of 16 with
# the threshold out of reach, independently of the copy engine (HSA_ENABLE_SDMA=0: 4 of 6) and with no page-locked registration of the
# engine alive (the fixture below) -- profiles/r05_experiments.md section 3.  [MiB]; an explicit setting in the environment wins
# (tools/experiments.sh faulthunt 10 - GPU_PINNED_MIN_XFER_SIZE=1).
os.environ.setdefault("GPU_PINNED_MIN_XFER_SIZE", "1048576")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(autouse=True)
def _no_live_host_registrations():
    """After every test: the engine holds no page-locked registration of a caller array ("pin_host_arrays").  A registration that
    outlives the array it was made for lets the runtime treat whatever the process maps there next as page-locked memory --
    the signature of the sporadic "Memory access fault by GPU ... Write access to a read-only page" of round 4."""
    yield
    from noahmp_amd import abi
    lib = getattr(abi, "_lib", None)
    if lib is not None:
        n = lib.noahmp_hip_debug_live_host_registrations()
        if n:
            lib.noahmp_hip_set_option(b"pin_host_arrays", 0)          # do not let one leak fail every later test
        assert n == 0, "%d host arrays are still page-locked by the engine after this test" % n


@pytest.fixture(scope="session")
def tables():
    from noahmp_amd.tables import load_tables
    return load_tables("usgs")          # (ctypes struct, dict)


@pytest.fixture(scope="session")
def port(tables):
    """The C restatement (oracle/_build), built on demand with gcc."""
    from oracle.portlib import PortLib
    p = PortLib(autobuild=True)
    p.set_tables(tables[0])
    return p


@pytest.fixture(scope="session")
def reflib():
    """The compiled reference (oracle/_ref); present only where `make -C oracle ref` has run."""
    from oracle import reflib as r
    if not r.available("O0"):
        pytest.skip("oracle/_ref/libnoahmp_ref.so not built (needs /root/reference)")
    return r.RefLib("O0")


@pytest.fixture(scope="session")
def engine(tables):
    from noahmp_amd.driver import Engine
    return Engine(tables[0], device=0)


def load_store(npz, prefix, ni, nj, cfg=None):
    """Rebuild a ColumnStore from a golden npz group."""
    from noahmp_amd.state import ColumnStore, ModelConfig
    s = ColumnStore(ni, nj, cfg or ModelConfig())
    for k in s.a:
        key = "%s/%s" % (prefix, k)
        if key in npz:
            s.a[k][...] = npz[key]
    return s
