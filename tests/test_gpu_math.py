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


def _pairs_as_doubles(n, d):
    """(n, d) float32 pairs packed as the low / high word of a float64 (what op 7 of cpol_debug_math unpacks)."""
    w = np.empty((len(n), 2), dtype=np.uint32)
    w[:, 0] = np.asarray(n, dtype=np.float32).view(np.uint32)
    w[:, 1] = np.asarray(d, dtype=np.float32).view(np.uint32)
    return w.view(np.float64).reshape(-1)


def test_float32_division_as_a_float64_product_has_the_bits_of_the_division(ctx):
    """The gate kernel forms its IEEE float32 quotients -- the vertical interpolation's (v2 - v1) / (z1 - z2), the cell
    coordinates (c - llc) / res (interpolation_c.c:43-57, 151) -- as RN32(RN64(n * r)), r = 1 / d to one float64 ulp
    (cpol_interp.inl: rcp_for_div32 / div32_by; the argument why that IS the correctly rounded quotient stands there).  Here
    the device compares the two forms bit for bit on 10^8 operand pairs: random bit patterns (every class: zeros, subnormals,
    infinities, NaN, both signs, all exponents), the magnitudes the kernel sees, quotients at and around the float32
    extremes, and every pair of a list of special values."""
    rng = np.random.default_rng(2026)
    bad = total = 0
    # (a) every pair of special values
    sp = np.array([0.0, -0.0, np.inf, -np.inf, np.nan, 1.0, -1.0, 3.0, 1.0 / 3.0, 1e-45, -1e-45, 1.1754942e-38, 1.17549435e-38,
                   3.4028235e38, -3.4028235e38, 1e-30, 1e30, 16777216.0, 16777215.0, 0.99999994, 1.0000001, 5e-324, 2.5e-39, 7e-46],
                  dtype=np.float32)
    nn, dd = np.meshgrid(sp, sp, indexing='ij')
    y = ctx.debug_math(7, _pairs_as_doubles(nn.ravel(), dd.ravel()))
    assert not y.any(), [(float(a), float(b)) for a, b, f in zip(nn.ravel(), dd.ravel(), y) if f][:8]
    total += y.size
    chunk = 10_000_000
    for k in range(10):
        if k < 4:        # random bit patterns
            n = rng.integers(0, 2 ** 32, chunk, dtype=np.uint64).astype(np.uint32).view(np.float32)
            d = rng.integers(0, 2 ** 32, chunk, dtype=np.uint64).astype(np.uint32).view(np.float32)
        elif k < 7:      # the kernel's operands: level thicknesses of 1 m .. 3 km, differences of anything from 1e-45 to 1e5
            d = rng.uniform(1.0, 3000.0, chunk).astype(np.float32) * rng.choice(np.array([1.0, -1.0], dtype=np.float32), chunk)
            n = (rng.standard_normal(chunk) * 10.0 ** rng.uniform(-45.0, 5.0, chunk)).astype(np.float32)
        elif k < 9:      # quotients near the float32 extremes and deep in the subnormals: exponents of n and d far apart
            n = (rng.uniform(1.0, 2.0, chunk) * 2.0 ** rng.integers(-149, 128, chunk)).astype(np.float32)
            d = (rng.uniform(1.0, 2.0, chunk) * 2.0 ** rng.integers(-149, 128, chunk)).astype(np.float32)
        else:            # significands that make quotients land next to rounding boundaries: small integers and their neighbours
            n = rng.integers(1, 2 ** 24, chunk).astype(np.float32)
            d = rng.integers(1, 2 ** 12, chunk).astype(np.float32) * np.float32(2.0) ** rng.integers(-20, 20, chunk).astype(np.float32)
        y = ctx.debug_math(7, _pairs_as_doubles(n, d))
        nb = int(y.sum())
        if nb:
            i = np.nonzero(y)[0][:5]
            raise AssertionError('chunk %d: %d quotients differ, e.g. %s' % (k, nb, [(n[j].item(), d[j].item()) for j in i]))
        total += chunk
    assert bad == 0 and total > 100_000_000
