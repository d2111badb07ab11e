"""Non-smooth scattering tables for the parity tests of the integral tables.

The synthetic tables of cosmo_pol_amd/synthetic.py are smooth in the diameter (Rayleigh spheroids
with a +-5 % modulation).  Real T-matrix tables at Ku / Ka band are not: resonance oscillations in
D, columns that change sign, columns whose PSD integral nearly cancels.  The integral tables
 interpolate the PSD-integrated result in the slope parameter lambda; what
they must survive is a table that is ROUGH IN D.  `roughen` derives such tables from a smooth one
(same layout, axes and magnitude envelope, so the observables stay finite)."""
import copy

import numpy as np

KINDS = ('random_sign', 'resonance', 'cancelling', 'alternating', 'mie')


def roughen(lut, kind, seed):
    tab = np.array(lut.value_table, dtype=np.float64, copy=True)       # [n_e, n_t, n_d, 12]
    n_e, n_t, n_d, n_c = tab.shape
    rng = np.random.default_rng(seed)
    k = np.arange(n_d, dtype=np.float64)
    if kind == 'random_sign':
        # every (slice, bin, column) entry scaled by an independent uniform factor in (-1, 1):
        # white noise in D with sign changes, nothing smooth left but the envelope
        tab *= rng.uniform(-1.0, 1.0, size=tab.shape)
    elif kind == 'resonance':
        # Mie-like ripples: a different period (7 .. 90 bins) and phase per column and slice,
        # 90 % modulation depth, plus a sharp resonance spike
        period = rng.uniform(7.0, 90.0, size=(n_e, n_t, 1, n_c))
        phase = rng.uniform(0.0, 2 * np.pi, size=(n_e, n_t, 1, n_c))
        tab *= 1.0 + 0.9 * np.sin(2 * np.pi * k[None, None, :, None] / period + phase)
        centre = rng.uniform(0.1, 0.9, size=(n_e, n_t, 1, n_c)) * n_d
        tab *= 1.0 + 6.0 / (1.0 + ((k[None, None, :, None] - centre) / 3.0) ** 2)
    elif kind in ('cancelling', 'alternating'):
        # sign flips from bin to bin in the odd columns: the PSD integral of those columns is a small
        # difference of two large sums for every lambda.  'cancelling': odd bins x -0.998, the integral
        # is ~1e-3 of the sum of |terms|.  'alternating': odd bins x -1, only boundary terms survive:
        # cancellation up to 1e10, where the 1024-term float64 sum itself (reference and device alike)
        # is rounding noise at the 1e-6 level -- the accuracy gate of the integral tables must notice
        # and leave such lambda ranges to the integrating kernels.
        # Columns 5 and 6 get opposite signs so that sz5 - sz6 and sz6 - sz5 do not vanish.
        alt = np.where(k % 2 == 0, 1.0, -0.998 if kind == 'cancelling' else -1.0)[None, None, :]
        for c in (1, 2, 5, 9, 11):
            tab[..., c] *= alt
        tab[..., 6] *= -alt
    else:
        raise ValueError(kind)
    out = copy.copy(lut)
    out.value_table = np.ascontiguousarray(tab)
    return out


def roughen_all(luts, kind, seed=20261004, frequency=None, scheme='1mom'):
    """`kind` 'mie': the non-melting species get closed-form Mie tables (cosmo_pol_amd/mie.py: real resonances
    along D at the case's frequency -- rain and hail at Ku / Ka band oscillate and change sign) on the axes of
    their smooth tables; the melting species keep theirs."""
    if kind == 'mie':
        from cosmo_pol_amd import mie
        return {h: (l if h in ('mS', 'mG') else mie.mie_table_like(l, h, frequency, scheme)) for h, l in luts.items()}
    return {h: roughen(l, kind, seed + 97 * j) for j, (h, l) in enumerate(sorted(luts.items()))}
