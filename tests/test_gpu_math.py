"""Accuracy of the device math helpers of the melting-species PSD kernel
(cosmo_pol_amd/csrc/cpol_psd.inl: cp_exp, cp_log, cp_cbrt, cp_cbrt_and_sixth,
cp_fourth_root, cp_rcp) against NumPy float64, over the argument ranges the kernel
produces.  Tolerance 2e-15 relative (a few ulp): the forward difference
(D_r(D + 0.01) - D_r(D)) / 0.01 amplifies cube-root errors by up to D / 0.01 ~ 1e3."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def ctx():
    from cosmo_pol_amd import _native
    c = _native.Context(0)
    yield c
    c.close()


def _rel(a, b):
    return np.max(np.abs(a - b) / np.abs(b))


def test_device_math_helpers(ctx):
    rng = np.random.default_rng(11)
    x = np.concatenate([-np.logspace(-6, np.log10(600.), 4000), np.logspace(-6, 2, 1000), [0.0]])
    y = ctx.debug_math(0, x)
    ref = np.exp(x)
    ok = ref > 1e-300
    assert _rel(y[ok], ref[ok]) < 2e-15
    pos = np.concatenate([np.logspace(-4, 4, 6000), rng.uniform(0.05, 12., 4000), [1.0, 2.0, 0.5]])
    l = ctx.debug_math(1, pos)
    refl = np.log(pos)
    assert np.max(np.abs(l - refl) / np.maximum(np.abs(refl), 1e-3)) < 2e-15
    assert _rel(ctx.debug_math(2, pos), np.cbrt(pos)) < 2e-15
    assert _rel(ctx.debug_math(3, pos), np.cbrt(pos)) < 2e-15
    assert _rel(ctx.debug_math(4, pos), pos ** (1.0 / 6.0)) < 2e-15
    assert _rel(ctx.debug_math(5, pos), np.sqrt(np.sqrt(pos))) < 2e-15
    assert _rel(ctx.debug_math(6, pos), 1.0 / pos) < 2e-15
