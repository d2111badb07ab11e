"""Seeded synthetic inputs: COSMO-like model cube and analytic scattering LUTs.

The reference ships neither model files nor lookup tables
(reference .gitignore:3 excludes cosmo_pol/lookup/lut_*/*; pytmatrix, needed by
cosmo_pol/lookup/compute_lut_sz.py to regenerate them, is not installable
here), so benches and tests use these generators.  The LUT generator produces
tables with the *exact layout* of the reference's tables
(compute_lut_sz.py:60-69: e = 0..90 step 2, t = 262..314 / 200..276 step 2,
wc = linspace(1e-3, 0.999, 100), 1024 diameters, 12 columns ordered as
compute_lut_sz.py:265-297) filled with a smooth Rayleigh-spheroid model plus a
seeded +-5 % modulation; it is NOT a T-matrix solver (table *values* are
"parity unpinned").
"""
import numpy as np

from . import constants as K
from .lut import Lookup_table

ELEVATIONS = np.arange(0, 91, 2)
TEMPERATURES_LIQ = np.arange(262, 316, 2)
TEMPERATURES_SOL = np.arange(200, 278, 2)
W_CONTENTS = np.linspace(1E-3, 0.999, 100)
NUM_DIAMETERS = 1024

_HYD_SEED = {'R': 1, 'S': 2, 'G': 3, 'H': 4, 'I': 5, 'mS': 6, 'mG': 7}


def _eps_water(T, f_ghz):
    """Single-Debye liquid water permittivity (smooth in T); complex."""
    tc = np.asarray(T, dtype=np.float64) - 273.15
    eps_s = 87.9 - 0.404 * tc + 9.59e-4 * tc ** 2
    eps_inf = 5.5
    f_rel = 9.0 + 0.4 * tc + 0.005 * tc ** 2        # GHz, increases with T
    x = f_ghz / np.maximum(f_rel, 1.0)
    return eps_inf + (eps_s - eps_inf) / (1 + x * x) + 1j * (eps_s - eps_inf) * x / (1 + x * x)


def _eps_ice_mix(rho_rel):
    """Ice/air mixture, rho_rel = particle density / ice density (clipped)."""
    r = np.clip(rho_rel, 0.005, 1.0)
    eps_i = 3.17 + 0.004j
    f = (eps_i - 1) / (eps_i + 2) * r              # Maxwell-Garnett, air matrix
    return (1 + 2 * f) / (1 - f)


def _spheroid_amplitudes(D, ar, eps, k0):
    """Rayleigh amplitudes (mm) along the horizontal (a) and the symmetry (z)
    axis of an oblate spheroid of equal-volume diameter D (mm)."""
    q = np.clip(ar, 0.05, 0.999)
    f2 = 1.0 / (q * q) - 1.0
    f = np.sqrt(f2)
    Lz = (1 + f2) / f2 * (1 - np.arctan(f) / f)
    Lx = (1 - Lz) / 2
    V = np.pi / 6 * D ** 3
    pol = lambda L: V / (4 * np.pi) * (eps - 1) / (1 + L * (eps - 1))
    return k0 * k0 * pol(Lx), k0 * k0 * pol(Lz)


def _modulation(rng, shape_et, n_d):
    """Smooth seeded factor 1 +- 5 % over (e, t, d)."""
    ne, nt = shape_et
    a = rng.uniform(0.3, 1.0, size=4)
    ph = rng.uniform(0, 2 * np.pi, size=4)
    e = np.linspace(0, 1, ne)[:, None, None]
    t = np.linspace(0, 1, nt)[None, :, None]
    d = np.linspace(0, 1, n_d)[None, None, :]
    m = (a[0] * np.sin(2 * np.pi * e + ph[0]) + a[1] * np.sin(3 * np.pi * t + ph[1])
         + a[2] * np.sin(5 * np.pi * d + ph[2]) + a[3] * np.sin(2 * np.pi * (e + t + d) + ph[3]))
    return 1.0 + 0.05 * m / np.sum(a)


def _axis_ratio(h, D, wc=None):
    if h == 'R':
        ar = 1.0048 + 5.7e-4 * D - 2.628e-2 * D ** 2 + 3.682e-3 * D ** 3 - 1.677e-4 * D ** 4
        return np.clip(ar, 0.45, 0.995)
    if h == 'S':
        return np.clip(0.85 - 0.008 * D, 0.55, 0.9)
    if h == 'G':
        return np.clip(0.93 - 0.004 * D, 0.8, 0.95)
    if h == 'H':
        return np.clip(0.97 - 0.012 * D, 0.75, 0.98)
    if h == 'I':
        return np.clip(0.5 - 0.12 * D, 0.15, 0.6)
    raise ValueError(h)


def _mass_params(h, scheme):
    c = K.C1 if scheme == '1mom' else K.C2
    if h == 'H':
        c = K.C2
    return getattr(c, 'AM_' + h), getattr(c, 'BM_' + h), getattr(c, 'D_MIN_' + h), getattr(c, 'D_MAX_' + h)


_RHO_CORR = {'R': 0.992, 'S': 0.93, 'G': 0.96, 'H': 0.95, 'I': 0.97, 'mS': 0.95, 'mG': 0.97}


def table_columns(h, Shh_b, Svv_b, Shh_f, Svv_f):
    """The 12 table columns (compute_lut_sz.py:265-297: Z11, Z12, Z21, Z22, Z33, Z34, Z43, Z44 of the
    back-scattering Mueller matrix, Re / Im S11, Re / Im S22 forward) from the co-polar amplitudes
    [n_e, n_t, n_d] (mm) of a particle without cross-polar scattering."""
    S11 = -Svv_b                                            # FSA sign in backscatter
    S22 = Shh_b + 0 * Svv_b
    rho0 = _RHO_CORR[h]
    shape = np.broadcast(S11, S22).shape
    tab = np.zeros(shape + (12,), dtype=np.float64)
    a11, a22 = np.abs(S11) ** 2, np.abs(S22) ** 2
    cross = S11 * np.conj(S22)
    tab[..., 0] = 0.5 * (a11 + a22)
    tab[..., 1] = 0.5 * (a11 - a22)
    tab[..., 2] = 0.5 * (a11 - a22)
    tab[..., 3] = 0.5 * (a11 + a22)
    tab[..., 4] = rho0 * cross.real
    tab[..., 5] = rho0 * cross.imag
    tab[..., 6] = -rho0 * cross.imag
    tab[..., 7] = rho0 * cross.real
    tab[..., 8] = np.broadcast_to(Svv_f.real, shape)
    tab[..., 9] = np.broadcast_to(Svv_f.imag, shape)
    tab[..., 10] = np.broadcast_to(Shh_f.real, shape)
    tab[..., 11] = np.broadcast_to(Shh_f.imag, shape)
    return tab


def _fill_table(h, D, eps, ar, k0, rng_mod):
    """D, eps, ar broadcastable to [ne, nt, nd] (e axis added here)."""
    Sa, Sz = _spheroid_amplitudes(D, ar, eps, k0)          # [1|., nt, nd]
    e = np.deg2rad(ELEVATIONS.astype(np.float64))[:, None, None]
    x = k0 * D / 2.0
    res = (1 - 0.08 * x * x) * np.exp(1j * 0.35 * x * x)    # mild resonance + phase
    Shh_b = Sa * res
    Svv_b = (Sa * np.sin(e) ** 2 + Sz * np.cos(e) ** 2) * res * (1 + 0.02j * x)
    Shh_f = Sa + 0j * e
    Svv_f = Sa * np.sin(e) ** 2 + Sz * np.cos(e) ** 2
    ext = lambda S: S.real + 1j * (S.imag + 2.0 / 3.0 * k0 * np.abs(S) ** 2)
    Shh_f, Svv_f = ext(Shh_f), ext(Svv_f)
    tab = table_columns(h, Shh_b, Svv_b, Shh_f, Svv_f)
    shape = tab.shape[:-1]
    mod = _modulation(rng_mod, shape[:2], shape[2])
    # the same factor on all 12 columns keeps the matrix physically consistent
    tab *= mod[..., None]
    return tab


def make_lut(h, frequency=5.6, scheme='1mom', seed=20260301, n_e=None, n_t=None):
    """Synthetic Lookup_table for hydrometeor `h` in the reference layout.
    n_e / n_t truncate the elevation / second axis (small test tables)."""
    wavelength = K.C_LIGHT / (frequency * 1e9) * 1000.0
    k0 = 2 * np.pi / wavelength
    rng = np.random.default_rng(seed + 1000 * _HYD_SEED[h] + int(round(frequency * 10)))
    elev = ELEVATIONS if n_e is None else ELEVATIONS[:n_e]
    if h in ('mS', 'mG'):
        solid = 'S' if h == 'mS' else 'G'
        wcs = W_CONTENTS if n_t is None else W_CONTENTS[:n_t]
        a_s, b_s, dmin_s, dmax_s = _mass_params(solid, scheme)
        a_r, b_r, dmin_r, dmax_r = _mass_params('R', scheme)
        fw = wcs[:, None]
        d_min = fw * dmin_r + (1 - fw) * dmin_s
        d_max = fw * dmax_r + (1 - fw) * dmax_s
        array_D = np.stack([np.linspace(d_min[i, 0], d_max[i, 0], NUM_DIAMETERS).astype('float32')
                            for i in range(len(wcs))])
        D = array_D.astype(np.float64)[None, :, :]
        m = fw[None] ** 2 * a_r * D ** b_r + (1 - fw[None] ** 2) * a_s * D ** b_s
        rho_rel = m / (np.pi / 6 * D ** 3) / K.RHO_I
        eps_dry = _eps_ice_mix(rho_rel)
        eps_w = _eps_water(273.15, frequency)
        eps = eps_dry + (eps_w - eps_dry) * fw[None] ** 1.5
        ar = (1 - fw[None]) * _axis_ratio(solid, D) + fw[None] * _axis_ratio('R', np.minimum(D, 8.0))
        tab = _fill_table(h, D, eps, ar, k0, rng)
        tab = tab[:len(elev)] if tab.shape[0] != len(elev) else tab
        axes = [('e', elev), ('wc', wcs), ('d', array_D)]
    else:
        temps = TEMPERATURES_LIQ if h == 'R' else TEMPERATURES_SOL
        if n_t is not None:
            temps = temps[:n_t]
        a_m, b_m, dmin, dmax = _mass_params(h, scheme)
        list_D = np.linspace(dmin, dmax, NUM_DIAMETERS).astype('float32')
        D = list_D.astype(np.float64)[None, None, :]
        T = temps.astype(np.float64)[None, :, None]
        if h == 'R':
            eps = _eps_water(T, frequency) + 0 * D
        else:
            rho_rel = a_m * D ** b_m / (np.pi / 6 * D ** 3) / K.RHO_I
            eps = _eps_ice_mix(rho_rel) * (1 + 2e-4 * (T - 240.0))
        ar = _axis_ratio(h, D) + 0 * T
        tab = _fill_table(h, D, eps, ar, k0, rng)
        tab = tab[:len(elev)]
        axes = [('e', elev), ('t', temps), ('d', list_D)]
    return Lookup_table.from_axes(axes + [('sz', np.arange(12))], np.ascontiguousarray(tab))


def make_all_luts(hydrometeors, frequency=5.6, scheme='1mom', seed=20260301, n_e=None, n_t=None):
    return {h: make_lut(h, frequency, scheme, seed, n_e, n_t) for h in hydrometeors}


# --------------------------------------------------------------------------
# synthetic model cube
# --------------------------------------------------------------------------

BENCH_GRID = dict(nz=80, ny=774, nx=1158, res=0.01, llc=(-6.8, -4.4), south_pole=(-43.0, 10.0))


def _bumps(rng, yy, xx, n, amp_lo, amp_hi, sig_lo, sig_hi):
    out = np.zeros(np.broadcast(yy, xx).shape, dtype=np.float32)
    y0, y1 = float(yy.min()), float(yy.max())
    x0, x1 = float(xx.min()), float(xx.max())
    for _ in range(n):
        cy, cx = rng.uniform(y0, y1), rng.uniform(x0, x1)
        s = rng.uniform(sig_lo, sig_hi)
        a = rng.uniform(amp_lo, amp_hi)
        out += (a * np.exp(-(((yy - cy) ** 2 + (xx - cx) ** 2) / (2 * s * s)))).astype(np.float32)
    return out


def make_cube(nz=80, ny=774, nx=1158, res=0.01, llc=(-6.8, -4.4), south_pole=(-43.0, 10.0),
              seed=20260301, two_moment=False, hydrometeors=('R', 'S', 'G', 'I'),
              model_top=22000.0, dense=True):
    """Returns dict(data={name: [nz,ny,nx] f32}, zlevels=[nz,ny,nx] f32,
    proj_info={...}, resolution=(dlon, dlat)).  Level 0 = model top.
    Fields: U,V,W,QR_v,QS_v,QG_v,QI_v,RHO,T (+ QH_v, QN*_v when two_moment).
    `dense`: stratiform precipitation everywhere (non-trivial but mostly
    positive QM>0 masks, the bench case)."""
    rng = np.random.default_rng(seed)
    rlat = (llc[1] + res * np.arange(ny, dtype=np.float64)).astype(np.float32)[:, None]
    rlon = (llc[0] + res * np.arange(nx, dtype=np.float64)).astype(np.float32)[None, :]
    span = max(ny, nx) * res
    topo = _bumps(rng, rlat, rlon, 6, 300.0, 1500.0, 0.08 * span, 0.25 * span)
    topo = np.clip(topo, 0.0, 3000.0).astype(np.float32)
    eta = ((nz - np.arange(nz) - 0.5) / nz).astype(np.float64) ** 1.5
    z = (topo[None] + (np.float32(model_top) - topo)[None] * eta.astype(np.float32)[:, None, None])
    z = z.astype(np.float32)

    tnoise = _bumps(rng, rlat, rlon, 8, -1.0, 1.0, 0.1 * span, 0.3 * span)
    T = (288.0 - 6.5e-3 * z + tnoise[None]).astype(np.float32)
    T = np.maximum(T, 205.0).astype(np.float32)
    RHO = (1.2 * np.exp(-z / 8000.0)).astype(np.float32)
    U = (8.0 + 6e-4 * z + 2.0 * tnoise[None]).astype(np.float32)
    V = (-3.0 + 2e-4 * z - 1.5 * tnoise[None]).astype(np.float32)
    cells = _bumps(rng, rlat, rlon, 12, 0.5, 1.0, 0.03 * span, 0.08 * span)
    cells = np.clip(cells, 0.0, 1.0).astype(np.float32)
    W = (cells[None] * 4.0 * np.sin(np.pi * np.clip(z / 12000.0, 0, 1))).astype(np.float32)

    strat = np.float32(0.35 if dense else 0.0) + np.float32(0.15) * np.clip(
        _bumps(rng, rlat, rlon, 5, 0.3, 1.0, 0.15 * span, 0.4 * span), 0, 1)
    strat = strat.astype(np.float32)
    inten = np.clip(strat + cells, 0.0, 1.3).astype(np.float32)[None]
    tc = T - np.float32(273.15)
    z0 = (288.0 - 273.15) / 6.5e-3                          # nominal freezing level [m]

    def ramp(v, lo, hi):
        return np.clip((v - lo) / (hi - lo), 0.0, 1.0).astype(np.float32)

    # rain below the 0C level (plus a 400 m overlap above it -> melting layer)
    QR = (2e-3 * inten * ramp(tc, -2.5, 2.0) * (0.6 + 0.4 * ramp(z, 0, 3000.0))).astype(np.float32)
    # snow above 0C-500 m
    QS = (1e-3 * inten * ramp(-tc, -3.0, 4.0) * ramp(model_top * 0.55 - z, 0, 3000.0)).astype(np.float32)
    # graupel in cells between 2 and 8 km
    QG = (1.5e-3 * cells[None] * ramp(z, 2000.0, 3200.0) * ramp(8000.0 - z, 0, 1500.0)).astype(np.float32)
    QG[QG < 2e-5] = 0.0
    # ice crystals colder than -15C
    QI = (1e-4 * inten * ramp(-tc, 15.0, 30.0) * ramp(model_top * 0.8 - z, 0, 3000.0)).astype(np.float32)
    QR[QR < 1e-6] = 0.0
    QS[QS < 1e-6] = 0.0
    QI[QI < 1e-7] = 0.0
    del z0
    data = {'U': U, 'V': V, 'W': W, 'QR_v': QR, 'QS_v': QS, 'QG_v': QG, 'QI_v': QI,
            'RHO': RHO, 'T': T}
    if two_moment:
        QH = (8e-4 * np.clip(cells - 0.7, 0, 1)[None] / 0.3 * ramp(z, 1000.0, 2500.0)
              * ramp(9000.0 - z, 0, 2000.0)).astype(np.float32)
        QH[QH < 2e-5] = 0.0
        data['QH_v'] = QH
        lim = {'R': (K.C2.X_MIN_R, K.C2.X_MAX_R), 'S': (K.C2.X_MIN_S, K.C2.X_MAX_S),
               'G': (K.C2.X_MIN_G, K.C2.X_MAX_G), 'H': (K.C2.X_MIN_H, K.C2.X_MAX_H),
               'I': (K.C2.X_MIN_I, K.C2.X_MAX_I)}
        for h in ('H', 'R', 'S', 'G', 'I'):
            lo, hi = np.log(lim[h][0] * 4), np.log(lim[h][1] / 4)
            xbar = np.exp(lo + (hi - lo) * (0.5 + 0.4 * np.sin(3.0 * tnoise + _HYD_SEED[h])))
            q = data['Q' + h + '_v']
            data['QN' + h + '_v'] = (q / xbar[None].astype(np.float32)).astype(np.float32)
    keep = {'R': 'QR_v', 'S': 'QS_v', 'G': 'QG_v', 'I': 'QI_v'}
    for h, name in keep.items():
        if h not in hydrometeors:
            data[name][:] = 0.0
    proj_info = {'Lo1': llc[0], 'La1': llc[1],
                 'Lo2': llc[0] + res * (nx - 1), 'La2': llc[1] + res * (ny - 1),
                 'Latitude_of_southern_pole': south_pole[0],
                 'Longitude_of_southern_pole': south_pole[1]}
    return dict(data=data, zlevels=z, proj_info=proj_info,
                resolution=np.asarray([res, res], dtype=np.float32))


def small_test_cube(center_rot=(-0.4725, -1.7207), half_width_deg=0.55, res=0.02, nz=30,
                    seed=7, **kw):
    """A small cube centred (in rotated coordinates) on the bench radar site,
    big enough for ~50 km range tests."""
    n = int(round(2 * half_width_deg / res)) + 1
    llc = (round(center_rot[1] - half_width_deg, 4), round(center_rot[0] - half_width_deg, 4))
    return make_cube(nz=nz, ny=n, nx=n, res=res, llc=llc, seed=seed, **kw)
