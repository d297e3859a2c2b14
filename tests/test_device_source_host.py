"""CPU check of the *device source*: noahmp_amd/csrc/*.hpp is __host__ __device__, so the very same
column_step() the GPU kernel runs is compiled here for the host (tests/host_emul, hipcc host pass,
host libm) and must reproduce the oracle BIT-FOR-BIT.  This isolates restructuring errors in the
GPU code from libm/ulp effects, without a GPU.  It is test infrastructure only: the product
library has no host path."""
import shutil

import numpy as np
import pytest

from noahmp_amd import synth
from noahmp_amd.abi import FIELD_INFO
from noahmp_amd.state import ModelConfig

pytestmark = pytest.mark.skipif(shutil.which("hipcc") is None and not __import__("os").path.exists("/opt/rocm/bin/hipcc"),
                                reason="hipcc not available")

SWEEP = [dict(), dict(idveg=1), dict(idveg=2), dict(idveg=5), dict(iopt_crs=2), dict(iopt_btr=2),
         dict(iopt_btr=3), dict(iopt_run=2), dict(iopt_run=3), dict(iopt_run=4), dict(iopt_run=5),
         dict(iopt_sfc=2), dict(iopt_frz=2), dict(iopt_inf=2), dict(iopt_rad=1), dict(iopt_rad=2),
         dict(iopt_alb=1), dict(iopt_snf=2), dict(iopt_snf=3), dict(iopt_tbot=1), dict(iopt_stc=2)]


@pytest.fixture(scope="module")
def emul(tables):
    from tests.host_emul.emullib import EmulLib
    e = EmulLib()
    e.set_tables(tables[0])
    return e


def _assert_same(a, b, what):
    for k in a.a:
        if FIELD_INFO[k][2] != "in":
            np.testing.assert_array_equal(a.a[k], b.a[k], err_msg="%s %s" % (k, what))


def test_device_source_free_run_bit_exact(emul, port, tables):
    s = synth.mixed_small(tables[1], ni=64, nj=4, glacier_frac=0.05)
    synth.first_step_fixups(s)
    so, se = s.copy(), s.copy()
    for it in range(1, 25):
        for x in (so, se):
            synth.diurnal_forcing(x, (it - 1) % 24, t_offset=s.t_offset)
        a = port.noahmplsm(so, it, 2000, 180.0)
        b = emul.noahmplsm(se, it, 2000, 180.0)
        assert (a.code, a.n_land, a.n_glacier) == (b.code, b.n_land, b.n_glacier)
    _assert_same(so, se, "free run 24 steps")


@pytest.mark.parametrize("kw", SWEEP, ids=[repr(k) for k in SWEEP])
def test_device_source_option_sweep_bit_exact(emul, port, tables, kw):
    s = synth.mixed_small(tables[1], ni=32, nj=4, cfg=ModelConfig(**kw))
    synth.first_step_fixups(s)
    so, se = s.copy(), s.copy()
    for it in range(1, 4):
        for x in (so, se):
            synth.diurnal_forcing(x, 10 + it, t_offset=s.t_offset)
        port.noahmplsm(so, it, 2000, 180.0)
        emul.noahmplsm(se, it, 2000, 180.0)
    _assert_same(so, se, kw)
