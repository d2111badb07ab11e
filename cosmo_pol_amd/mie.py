"""Closed-form Mie scattering tables in the reference's table layout (test tables with REAL resonances).

The synthetic tables of synthetic.py are Rayleigh spheroids: smooth in the diameter.  T-matrix tables
(cosmo_pol/lookup/compute_lut_sz.py:60-69, 265-297; pytmatrix is absent here) are not -- at Ku and Ka
band rain and hail go through Mie resonances, columns oscillate and change sign along D.  This module
fills a table of the exact layout [n_e, n_t, 1024, 12] (same axes as the table it is given) from the
Lorenz-Mie series of the equal-volume sphere:

    a_n, b_n          Bohren & Huffman's algorithm (logarithmic derivative by downward recurrence,
                      Riccati-Bessel functions upward, series cut at n = x + 4 x^(1/3) + 2), NumPy only
    F(0), F(pi)       forward / backward amplitude in mm:  F = i S / k,
                      S(0) = 1/2 sum (2n+1)(a_n + b_n),  S(pi) = 1/2 sum (2n+1)(-1)^n (a_n - b_n)

A sphere has Z_DR = 1 and K_DP = 0, which would leave half of the 12 columns degenerate; the two
polarisations therefore carry the sphere's amplitude times the ratio (spheroid / sphere) of the
RAYLEIGH amplitudes of synthetic.py's axis-ratio model, seen under the table's elevation angle.  In
the Rayleigh limit the table equals the spheroid table (without its ad-hoc resonance factor and
modulation); beyond it the D-dependence is the sphere's Mie series.  A test table -- not a
substitute for T-matrix tables (*parity unpinned*, like every table value here).
"""
import numpy as np

from . import constants as K
from . import synthetic as syn
from .lut import Lookup_table


def mie_amplitudes(x, m):
    """Forward and backward scattering functions S(0), S(pi) of spheres of size parameter x (real array)
    and refractive index m (complex, broadcastable to x)."""
    x = np.asarray(x, dtype=np.float64)
    m = np.broadcast_to(np.asarray(m, dtype=np.complex128), x.shape)
    y = m * x
    nstop = np.floor(x + 4.0 * np.cbrt(x) + 2.0).astype(np.int64)
    nmax = int(nstop.max())
    nmx = int(max(nmax, np.ceil(np.abs(y).max()))) + 16
    # logarithmic derivative D_n(mx), downward
    D = np.zeros((nmx + 1,) + x.shape, dtype=np.complex128)
    for n in range(nmx, 0, -1):
        D[n - 1] = n / y - 1.0 / (D[n] + n / y)
    psi0, psi1 = np.cos(x), np.sin(x)
    chi0, chi1 = -np.sin(x), np.cos(x)
    xi1 = psi1 - 1j * chi1
    s_f = np.zeros(x.shape, dtype=np.complex128)
    s_b = np.zeros(x.shape, dtype=np.complex128)
    for n in range(1, nmax + 1):
        psi = (2 * n - 1) / x * psi1 - psi0
        chi = (2 * n - 1) / x * chi1 - chi0
        xi = psi - 1j * chi
        da = D[n] / m + n / x
        db = m * D[n] + n / x
        an = (da * psi - psi1) / (da * xi - xi1)
        bn = (db * psi - psi1) / (db * xi - xi1)
        use = n <= nstop                              # (beyond the series' end the upward recurrence is noise)
        s_f += np.where(use, 0.5 * (2 * n + 1) * (an + bn), 0.0)
        s_b += np.where(use, 0.5 * (2 * n + 1) * (-1) ** n * (an - bn), 0.0)
        psi0, psi1 = psi1, psi
        chi0, chi1 = chi1, chi
        xi1 = psi1 - 1j * chi1
    return s_f, s_b


def mie_table_like(lut, h, frequency, scheme='1mom'):
    """-> Lookup_table with the axes of `lut` (a table of a non-melting species h) and Mie values."""
    if h in ('mS', 'mG'):
        raise ValueError('Mie tables are made for the non-melting species')
    wavelength = K.C_LIGHT / (frequency * 1e9) * 1000.0
    k0 = 2 * np.pi / wavelength
    names = lut.axes_names
    elev = np.asarray(lut.axes[names['e']], dtype=np.float64)
    temps = np.asarray(lut.axes[names['t']], dtype=np.float64)
    list_D = np.asarray(lut.axes[names['d']], dtype=np.float32)
    D = list_D.astype(np.float64)[None, :]
    T = temps[:, None]
    if h == 'R':
        eps = syn._eps_water(T, frequency) + 0 * D
    else:
        a_m, b_m, _, _ = syn._mass_params(h, scheme)
        rho_rel = a_m * D ** b_m / (np.pi / 6 * D ** 3) / K.RHO_I
        eps = syn._eps_ice_mix(rho_rel) * (1 + 2e-4 * (T - 240.0))
    ar = syn._axis_ratio(h, D) + 0 * T
    s_f, s_b = mie_amplitudes(k0 * D / 2.0 + 0 * T, np.sqrt(eps))
    f_f, f_b = 1j * s_f / k0, 1j * s_b / k0                     # [n_t, n_d] amplitudes of the sphere, mm
    # polarisation split: Rayleigh spheroid over Rayleigh sphere, under the elevation angle
    sa, sz = syn._spheroid_amplitudes(D, ar, eps, k0)
    sphere = k0 * k0 * D ** 3 / 8.0 * (eps - 1) / (eps + 2)
    e = np.deg2rad(elev)[:, None, None]
    r_h = (sa / sphere)[None] + 0 * e
    r_v = (sa[None] * np.sin(e) ** 2 + sz[None] * np.cos(e) ** 2) / sphere[None]
    tab = syn.table_columns(h, f_b[None] * r_h, f_b[None] * r_v, f_f[None] * r_h, f_f[None] * r_v)
    return Lookup_table.from_axes([('e', np.asarray(lut.axes[names['e']])), ('t', np.asarray(lut.axes[names['t']])),
                                   ('d', list_D), ('sz', np.arange(12))], np.ascontiguousarray(tab))


def make_lut(h, frequency=5.6, scheme='1mom', n_e=None, n_t=None):
    """Mie table of species h on the reference's axes (those of synthetic.make_lut)."""
    return mie_table_like(syn.make_lut(h, frequency, scheme, n_e=n_e, n_t=n_t), h, frequency, scheme)
