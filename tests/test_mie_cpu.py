"""The closed-form Mie table generator (cosmo_pol_amd/mie.py): the Lorenz-Mie series against published
efficiencies and the Rayleigh limit; the table layout; resonances along D at Ka band."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cosmo_pol_amd import mie, synthetic  # noqa: E402


def test_mie_series_against_published_efficiencies():
    # Wiscombe (1979) MIEV0 test cases / Bohren & Huffman appendix A (BHMIE sample run)
    for x, m, qext, qback in ((10.0, 1.5, 2.881999, None), (1.0, 1.33, 0.093924, None),
                              (5.2128, 1.55, 3.10543, 2.92534)):
        sf, sb = mie.mie_amplitudes(np.array([x]), m)
        assert abs(4 / x ** 2 * sf.real[0] - qext) < 2e-5 * qext + 2e-6, (x, m)
        if qback is not None:
            assert abs(4 * abs(sb[0]) ** 2 / x ** 2 - qback) < 2e-4 * qback
    # Rayleigh limit: S(0) -> -i x^3 (m^2 - 1) / (m^2 + 2), S(pi) -> -S(0)
    m = 1.5 + 0.1j
    x = np.array([1e-3, 1e-2])
    sf, sb = mie.mie_amplitudes(x, m)
    ray = -1j * x ** 3 * (m * m - 1) / (m * m + 2)
    assert np.allclose(sf, ray, rtol=2e-4) and np.allclose(sb, -ray, rtol=2e-4)
    # vectorised over size parameter and refractive index alike
    xs = np.linspace(0.05, 20.0, 64)
    a, _ = mie.mie_amplitudes(xs, 1.5)
    b = np.array([mie.mie_amplitudes(np.array([v]), 1.5)[0][0] for v in xs])
    assert np.allclose(a, b, rtol=1e-12)


def test_mie_table_layout_rayleigh_limit_and_resonances():
    for h, freq, scheme in (('R', 5.6, '1mom'), ('H', 35.6, '2mom'), ('R', 35.6, '2mom')):
        smooth = synthetic.make_lut(h, freq, scheme, n_e=3, n_t=2)
        t = mie.mie_table_like(smooth, h, freq, scheme)
        assert t.value_table.shape == smooth.value_table.shape == (3, 2, 1024, 12)
        assert t.value_table.dtype == np.float64 and np.all(np.isfinite(t.value_table))
        for k in ('e', 't', 'd'):
            assert np.array_equal(t.axes[t.axes_names[k]], smooth.axes[smooth.axes_names[k]])
        z11 = t.value_table[0, 0, :, 0]
        assert np.all(z11 > 0)
    # small drops at C band: the Mie table is the Rayleigh-spheroid table (same polarisation split)
    smooth = synthetic.make_lut('R', 5.6, '1mom', n_e=2, n_t=2)
    t = mie.mie_table_like(smooth, 'R', 5.6, '1mom')
    d = np.asarray(t.axes[t.axes_names['d']], dtype=np.float64)
    small = d < 0.5
    ratio = t.value_table[0, 0, small, 10] / smooth.value_table[0, 0, small, 10]      # Re S_hh forward
    assert np.all(np.abs(ratio - 1) < 0.08)                                          # (+-5 % modulation of the smooth table)
    # hail at Ka band: Z11 oscillates along D (resonances), the smooth table does not
    sm = synthetic.make_lut('H', 35.6, '2mom', n_e=2, n_t=2)
    mt = mie.mie_table_like(sm, 'H', 35.6, '2mom')
    dz = np.diff(np.log(mt.value_table[0, 0, 50:, 0]))
    turns = int(np.sum(np.sign(dz[1:]) != np.sign(dz[:-1])))
    dz_s = np.diff(np.log(sm.value_table[0, 0, 50:, 0]))
    assert turns >= 6 and int(np.sum(np.sign(dz_s[1:]) != np.sign(dz_s[:-1]))) <= 2, turns
