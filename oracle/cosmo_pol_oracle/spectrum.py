"""Oracle Doppler spectrum, Doppler scheme 3 (TEST INFRASTRUCTURE ONLY).

Restates, for hydrometeors with power-law fall speeds (R, S, G, H, I) and -- round 6 -- for the
melting species (mS, mG), whose fall speed is inverted through a linear interpolator over
(V(D_k), D_k) that the reference rebuilds at EVERY gate: `set_psd` clears `vd_interpolator`
(cosmo_pol/hydrometeors/hydrometeors.py:1426, 1474), `get_D_from_V` (:480-500) builds it from
the gate's wet fraction.  The reference's melting branch needs NumPy < 1.16 (`np.linspace` with
one-element array endpoints, doppler_scatter.py:687) and drops mS / mG from a sub-beam without
melting through `dict.keys().remove`, which raises under Python 3 (:343-349): the goldens
`radial_d3_melt*` come from the reference under oracle/ref_shim.py::numpy1_linspace on radials
whose sub-beams all cross the melting layer; a sub-beam without melting is restated as intended
(mS / mG skipped):
  get_diameter_from_rad_vel   cosmo_pol/scatter/doppler_scatter.py:545-600
  get_doppler_spectrum        :603-716
  get_refl                    cosmo_pol/scatter/doppler_c.c:11-32 (float32, sequential sums)
  per-sub-beam attenuation and accumulation            :297-305, 353-391
  spectral_width_motion / broaden_spectrum             :756-802
  RVEL from the spectrum                               :422-429
Quirks reproduced: the diameter clamp of hydrometeor j is applied to the WHOLE matrix
(columns of the hydrometeors before it are re-clamped with its limits, :583-586); the
radar constant uses K_squared SQUARED (:709); the per-hydrometeor attenuation of the
r-th VALID gate is added to gate r (zero padding of nansum_arr, :305) before the
cumulative sum; velocity-bin edges are the points of VARRAY where the inverted fall
speed is >= 0 and a row is kept when at least one hydrometeor has a non-empty bin.
"""
import numpy as np

from . import constants as K
from .beam import nansum_pair
from .psd import create_hydrometeor

F32 = np.float32


def velocity_array(config):
    """global_constants.py:168-171."""
    nvel = K.C_LIGHT / (4 * 1e-6 * config['radar']['PRI'] * config['radar']['frequency'] * 1e9)
    vres = 2 * nvel / config['radar']['FFT_length']
    return np.arange(-nvel, nvel + vres, vres)


def interp1d_linear(x, y, x_new):
    """scipy.interpolate.interp1d(x, y, kind='linear', bounds_error=False, fill_value=nan,
    assume_sorted=False) restated (what hydrometeors.py:494-500 builds and calls).  For 1-D float64 data
    SciPy (0.17 to 1.15 alike) sorts by x (stable) and hands the evaluation to `np.interp`
    (interp1d._call_linear_np), then overwrites the queries outside [x[0], x[-1]] with the fill value.
    np.interp (numpy/_core/src/multiarray/compiled_base.c::arr_interp): j with x[j] <= q < x[j+1] by binary
    search, y[j] when q == x[j] or j is the last node, else slope * (q - x[j]) + y[j] with
    slope = (y[j+1] - y[j]) / (x[j+1] - x[j]); a NaN query stays NaN."""
    x = np.asarray(x, dtype=np.float64)
    y = np.asarray(y, dtype=np.float64)
    order = np.argsort(x, kind='mergesort')
    x, y = x[order], y[order]
    q = np.asarray(x_new, dtype=np.float64)
    n = len(x)
    j = np.clip(np.searchsorted(x, q, side='right') - 1, 0, n - 1)
    jn = np.minimum(j + 1, n - 1)
    with np.errstate(invalid='ignore', divide='ignore'):
        slope = (y[jn] - y[j]) / (x[jn] - x[j])
        out = slope * (q - x[j]) + y[j]
        # "if we get nan in one direction, try the other" (compiled_base.c)
        other = slope * (q - x[jn]) + y[jn]
        out = np.where(np.isnan(out), np.where(np.isnan(other) & (y[j] == y[jn]), y[j], other), out)
    exact = (j == n - 1) | (x[j] == q)
    out = np.where(exact, y[j], out)
    out = np.where(np.isnan(q), q, out)
    out[(q < x[0]) | (q > x[-1])] = np.nan
    return out


def melting_diameter_from_velocity(h, wh):
    """get_D_from_V of a melting species (hydrometeors.py:480-500) for the wet fraction `h` holds now."""
    D_all = np.linspace(np.min(h.d_min), np.max(h.d_max), h.nbins_D)
    V_all = np.squeeze(h.get_V(D_all))
    return interp1d_linear(V_all, D_all, wh)


def diameters_from_radial_velocity(hyds, limits, varray, phi_deg, theta_deg, U, V, W, rho_corr):
    theta = np.deg2rad(theta_deg)
    phi = np.deg2rad(phi_deg)
    with np.errstate(invalid='ignore', divide='ignore'):
        wh = (1. / rho_corr * (W + (U * np.sin(phi) + V * np.cos(phi)) / np.tan(theta)
                               - varray / np.sin(theta)))
        idx = np.where(wh >= 0)[0]
        wh = wh[idx]
        D = np.zeros((len(idx), len(hyds)), dtype='float32')
        for i, h in enumerate(hyds):
            if hasattr(h, 'alpha'):
                D[:, i] = (wh / h.alpha) ** (1. / h.beta)
            else:                          # melting species: NaN outside the tabulated fall speeds stays NaN below
                D[:, i] = melting_diameter_from_velocity(h, wh)
            d_min, d_max = limits[i]
            D[D >= d_max] = d_max          # whole matrix (quirk)
            D[D <= d_min] = d_min
    Da = np.minimum(D[0:-1, :], D[1:, :])
    Db = np.maximum(D[0:-1, :], D[1:, :])
    mask = np.where(np.sum((Db - Da) == 0.0, axis=1) < len(hyds))[0]
    return Da[mask, :], Db[mask, :], idx[mask]


def bin_reflectivities(Da, Db, rcs, N, step_D, D_min):
    """doppler_c.c:11-32 in float32 with sequential accumulation."""
    n_rows, n_h = Da.shape
    refl = np.zeros(n_rows, dtype=F32)
    prod = (N.astype(F32) * rcs.astype(F32)).astype(F32)
    with np.errstate(invalid='ignore', over='ignore'):
        ia = ((Da - D_min[None, :]) / step_D[None, :]).astype(F32)
        ib = ((Db - D_min[None, :]) / step_D[None, :]).astype(F32)
        # (int) truncation; a NaN edge (melting species, fall speed outside the interpolator) converts to INT_MIN on
        # x86-64 for BOTH edges (np.minimum / np.maximum propagate it to Da and Db alike): an empty bin
        nan_edge = np.isnan(ia) | np.isnan(ib)
        ia = np.where(nan_edge, 0, ia).astype(np.int64)
        ib = np.where(nan_edge, 0, ib).astype(np.int64)
    for r in range(n_rows):
        for j in range(n_h):
            s = F32(0)
            if ib[r, j] > ia[r, j]:
                s = np.cumsum(prod[ia[r, j]:ib[r, j], j], dtype=F32)[-1]
            refl[r] = F32(refl[r] + F32(s * step_D[j]))
    return refl


def subbeam_spectrum(sb, hydro_names, luts, config, varray):
    """get_doppler_spectrum for one sub-beam -> float32 [n_gates, n_v]."""
    scheme = config['microphysics']['scheme']
    n_gates = len(sb.dist_profile)
    refl = np.zeros((n_gates, len(varray)), dtype='float32')
    elev = sb.elev_profile
    elev_lut = elev.copy()
    elev_lut[elev_lut > 90] = 180 - elev_lut[elev_lut > 90]
    elev_lut[elev_lut < 0] = -elev_lut[elev_lut < 0]
    phi = sb.quad_pt[0]
    wavelength = K.Derived(config).WAVELENGTH
    KW = config['radar']['K_squared']
    with np.errstate(invalid='ignore', divide='ignore'):
        rho_corr = (sb.values['RHO'] / sb.values['RHO'][0]) ** 0.5
    objs = {}
    if not getattr(sb, 'has_melting', True):
        # (intended behaviour of doppler_scatter.py:343-349; as written it raises under Python 3)
        hydro_names = [h for h in hydro_names if h not in ('mS', 'mG')]
    for h in hydro_names:
        objs[h] = create_hydrometeor(h, scheme)
        d_ax = luts[h].axes[2]
        objs[h].nbins_D = luts[h].value_table.shape[-2]
        if h not in ('mS', 'mG'):          # (melting: set by the wet fraction of the gate, hydrometeors.py:332-339)
            objs[h].d_min, objs[h].d_max = d_ax[0], d_ax[-1]
    for i in range(n_gates):
        if sb.mask[i] != 0:
            continue
        if not np.isscalar(sb.quad_weight) and sb.quad_weight[i] == 0:
            continue
        present = []
        T = sb.values['T'][i]
        for h in hydro_names:
            Q = sb.values['Q' + h + '_v'][i]
            if Q > 0:
                present.append(h)
                if scheme == '1mom':
                    if h == 'mG':
                        objs[h].set_psd(np.array([Q]), np.array([sb.values['fwet_' + h][i]]))
                    elif h == 'mS':
                        objs[h].set_psd(np.array([T]), np.array([Q]), np.array([sb.values['fwet_' + h][i]]))
                    elif h in ['S', 'I']:
                        objs[h].set_psd(np.array([T]), np.array([Q]))
                    else:
                        objs[h].set_psd(np.array([Q]))
                else:
                    objs[h].set_psd(np.array([sb.values['QN' + h + '_v'][i]]), np.array([Q]))
        n_h = len(present)
        n_d = luts[hydro_names[-1]].value_table.shape[-2]
        rcs = np.zeros((n_d, n_h), dtype='float32') + np.nan
        N = np.zeros((n_d, n_h), dtype='float32') + np.nan
        D = np.zeros((n_d, n_h), dtype='float32') + np.nan
        D_min = np.zeros((n_h), dtype='float32') + np.nan
        step_D = np.zeros((n_h), dtype='float32') + np.nan
        with np.errstate(invalid='ignore', over='ignore', divide='ignore'):
            for j, h in enumerate(present):
                # (melting: d_min / d_max have shape (1,); NumPy < 1.16's linspace gave shape (n_d,) for them)
                D[:, j] = np.linspace(np.ravel(objs[h].d_min)[0], np.ravel(objs[h].d_max)[0], objs[h].nbins_D)
                D_min[j] = D[0, j]
                step_D[j] = D[1, j] - D[0, j]
                N[:, j] = objs[h].get_N(D[:, j])
                if h in ('mS', 'mG'):
                    sz = luts[h].lookup_line(e=elev_lut[i], wc=sb.values['fwet_' + h][i])
                else:
                    sz = luts[h].lookup_line(e=elev_lut[i], t=T)
                rcs[:, j] = (2 * np.pi * (sz[:, 0] - sz[:, 1] - sz[:, 2] + sz[:, 3])).T
        Da, Db, idx = diameters_from_radial_velocity(
            [objs[h] for h in present], [(objs[h].d_min, objs[h].d_max) for h in present], varray,
            phi, elev[i], sb.values['U'][i], sb.values['V'][i], sb.values['W'][i], rho_corr[i])
        refl[i, idx] = bin_reflectivities(Da, Db, rcs, N, step_D, D_min)
        refl[i, idx] *= wavelength ** 4 / (np.pi ** 5 * KW ** 2)
    return refl


def spectral_width_motion(elevations, config):
    wavelength = K.Derived(config).WAVELENGTH / 100.
    return ((wavelength * config['radar']['antenna_speed'] * np.cos(np.deg2rad(elevations)))
            / (2 * np.pi * np.deg2rad(config['radar']['3dB_beamwidth'])))


def broaden_spectrum(spectrum, std, varray):
    from scipy.ndimage import gaussian_filter
    v_res = varray[2] - varray[1]
    original_power = np.sum(spectrum, 1)
    for i, t in enumerate(std):
        spectrum[i, :] = gaussian_filter(spectrum[i, :], t / v_res)
    convolved_power = np.sum(spectrum, 1)
    with np.errstate(invalid='ignore', divide='ignore'):
        return spectrum / convolved_power[:, None] * original_power[:, None]


def attenuation_per_beam(ah_list, n_gates):
    """ah_list: per hydrometeor, the [n_valid] float64 one-way attenuation x bin length of
    its VALID gates; summed front-aligned into a float32 [n_gates] vector (:156, :305)."""
    acc = np.zeros((n_gates,), dtype='float32') + np.nan
    for ah in ah_list:
        acc = nansum_pair(acc, ah)
    return acc


def apply_attenuation(beam_spectrum, ah_per_beam):
    """:372-384."""
    x = np.array(ah_per_beam, copy=True)
    x[np.isnan(x)] = 0
    ahc = np.cumsum(x)
    with np.errstate(invalid='ignore', divide='ignore'):
        sum_power = np.nansum(beam_spectrum, axis=1)
        idx_valid = sum_power > 0
        sum_power_db = 10 * np.log10(sum_power)
        frac = sum_power[idx_valid] / (10 ** (0.1 * (sum_power_db[idx_valid] - ahc[idx_valid])))
    beam_spectrum[idx_valid, :] /= frac[:, None]
    return beam_spectrum


def rvel_from_spectrum(spectrum, varray):
    n_gates = spectrum.shape[0]
    with np.errstate(invalid='ignore', divide='ignore'):
        rv = np.nansum(np.tile(varray, (n_gates, 1)) * spectrum, axis=1)
        rv /= np.nansum(spectrum, axis=1)
    return rv
