"""Oracle microphysics + scattering of one radial (TEST INFRASTRUCTURE ONLY).

Restates cosmo_pol/scatter/doppler_scatter.py for Doppler schemes 1/2 off or
scheme 1 (scheme 3 / Doppler spectrum is out of scope):
  get_radar_observables   :49-489   (PSD :184-211, diameters :216-230,
                                     LUT :236-241, rectangle-rule integral
                                     :246-251, accumulate :259-268,
                                     Doppler scheme 1 :276-281,313-333,418-437,
                                     final :400-416, mask :472-477)
  get_pol_from_sz         :491-544
  cut_at_sensitivity      :804-862
  nan_cumsum / nan_cumprod / aliasing : cosmo_pol/utilities/utilities.py:142-199
Quirks reproduced: Q3 (weights not renormalised), Q4 (float32 sz_integ), Q5
(exact zeros -> NaN), Q6 (sensitivity ranges from 0), Q7 (ZDR attenuated,
ZH/ZV not), Q8 (elevation folded in place).
"""
import numpy as np

from . import constants as K
from .beam import SubBeam, nansum_pair
from .config import hydrometeor_list
from .psd import create_hydrometeor, vlinspace


def nan_cumsum(x):
    x[np.isnan(x)] = 0
    return np.cumsum(x)


def nan_cumprod(x):
    x[np.isnan(x)] = 1
    return np.cumprod(x)


def proj_vel(U, V, W, vf, theta, phi):
    return ((U * np.sin(phi) + V * np.cos(phi)) * np.cos(theta) + (W - vf) * np.sin(theta))


def pol_from_sz(sz, config):
    """(z_h, z_v, zdr, rhohv, kdp, ah, av, delta_hv) from [n,12] integrated
    scattering entries (doppler_scatter.py:491-544)."""
    wavelength = K.Derived(config).WAVELENGTH
    K2 = config['radar']['K_squared']
    with np.errstate(invalid='ignore', divide='ignore'):
        xs_h = 2 * np.pi * (sz[:, 0] - sz[:, 1] - sz[:, 2] + sz[:, 3])
        z_h = wavelength ** 4 / (np.pi ** 5 * K2) * xs_h
        xs_v = 2 * np.pi * (sz[:, 0] + sz[:, 1] + sz[:, 2] + sz[:, 3])
        z_v = wavelength ** 4 / (np.pi ** 5 * K2) * xs_v
        zdr = xs_h / xs_v
        kdp = 1e-3 * (180.0 / np.pi) * wavelength * (sz[:, 10] - sz[:, 8])
        ah = 4.343e-3 * (2 * wavelength * sz[:, 11])
        av = 4.343e-3 * (2 * wavelength * sz[:, 9])
        a = (sz[:, 4] + sz[:, 7]) ** 2 + (sz[:, 6] - sz[:, 5]) ** 2
        b = (sz[:, 0] - sz[:, 1] - sz[:, 2] + sz[:, 3])
        c = (sz[:, 0] + sz[:, 1] + sz[:, 2] + sz[:, 3])
        rhohv = np.sqrt(a / (b * c))
        delta_hv = np.arctan2(sz[:, 5] - sz[:, 6], -sz[:, 4] - sz[:, 7])
    return z_h, z_v, zdr, rhohv, kdp, ah, av, delta_hv


def aliasing(v, nyquist):
    """Velocity folding into [-nyquist, nyquist] (utilities.py:142-156)."""
    theta = (v + nyquist) / (2 * nyquist) * np.pi - np.pi / 2.
    theta_fold = np.arctan(np.tan(theta))
    return (theta_fold + np.pi / 2) * (2 * nyquist) / np.pi - nyquist


def radar_observables(subbeams, luts, config, return_sz=False, doppler=True, nyquist=None):
    """One radial: list of SubBeam -> SubBeam of radar observables."""
    mp = config['microphysics']
    scheme = mp['scheme']
    melting = mp['with_melting']
    radial_res = config['radar']['radial_resolution']
    dop_scheme = config['doppler']['scheme']
    simulate_doppler = doppler and config['radar'].get('type', 'ground') != 'GPM' \
        and dop_scheme in (1, 2, 3)
    hydrom_types = hydrometeor_list(config)

    n_sub = len(subbeams)
    idx_0 = int(n_sub / 2)
    n_gates = max([len(sb.dist_profile) for sb in subbeams])
    if simulate_doppler and dop_scheme == 3:
        from . import spectrum as SP
        varray = SP.velocity_array(config)
        doppler_spectrum = np.zeros((n_gates, len(varray)))

    hyd = {}
    for h in hydrom_types:
        hyd[h] = create_hydrometeor(h, scheme)
        hyd[h].nbins_D = luts[h].value_table.shape[-2]
        d_ax = luts[h].axes[2]
        hyd[h].d_min = d_ax[:, 0] if h in ['mS', 'mG'] else d_ax[0]
        hyd[h].d_max = d_ax[:, -1] if h in ['mS', 'mG'] else d_ax[-1]

    # scheme 'ml': per-gate sub-beam weights, renormalised gate by gate (:124-129)
    array_weights = not np.isscalar(subbeams[0].quad_weight)
    if array_weights:
        total_weight_at_gates = np.sum(np.array([b.quad_weight for b in subbeams]), axis=0)

    sz_integ = np.zeros((n_gates, len(hydrom_types), 12), dtype='float32') + np.nan
    rvel_avg = np.zeros(n_gates,) + np.nan
    total_weight_rvel = np.zeros(n_gates,)

    with np.errstate(invalid='ignore', divide='ignore', over='ignore'):
        for sb in subbeams:
            v_integ = np.zeros(n_gates,)
            n_integ = np.zeros(n_gates,)
            ah_list = []
            for j, h in enumerate(hydrom_types):
                if melting and not sb.has_melting and h in ['mS', 'mG']:
                    continue
                elev_lut = sb.elev_profile          # in place (Q8)
                elev_lut[elev_lut > 90] = 180 - elev_lut[elev_lut > 90]
                elev_lut[elev_lut < 0] = - elev_lut[elev_lut < 0]
                T = sb.values['T']
                QM = sb.values['Q' + h + '_v']
                valid = QM > 0
                if array_weights:                    # :186-189
                    valid = np.logical_and(valid, sb.quad_weight > 0)
                if not np.any(valid):
                    continue
                if scheme == '1mom':
                    if h == 'mG':
                        fwet = sb.values['fwet_' + h]
                        hyd[h].set_psd(QM[valid], fwet[valid])
                    elif h == 'mS':
                        fwet = sb.values['fwet_' + h]
                        hyd[h].set_psd(T[valid], QM[valid], fwet[valid])
                    elif h in ['S', 'I']:
                        hyd[h].set_psd(T[valid], QM[valid])
                    else:
                        hyd[h].set_psd(QM[valid])
                else:
                    QN = sb.values['QN' + h + '_v']
                    hyd[h].set_psd(QN[valid], QM[valid])

                if h in ['mS', 'mG']:
                    list_D = vlinspace(hyd[h].d_min, hyd[h].d_max, hyd[h].nbins_D)
                    dD = list_D[:, 1] - list_D[:, 0]
                else:
                    list_D = luts[h].axes[luts[h].axes_names['d']]
                    dD = list_D[1] - list_D[0]
                N = hyd[h].get_N(list_D)
                if len(N.shape) == 1:
                    N = np.reshape(N, [len(N), 1])

                if h in ['mS', 'mG']:
                    sz = luts[h].lookup_line(e=elev_lut[valid], wc=fwet[valid])
                    sz_psd = np.einsum('ijk,ij->ik', sz, N) * dD[:, None]
                else:
                    sz = luts[h].lookup_line(e=elev_lut[valid], t=T[valid])
                    sz_psd = np.einsum('ijk,ij->ik', sz, N) * dD

                if array_weights:                    # :259-264
                    w = sb.quad_weight[valid] / total_weight_at_gates[valid]
                    sz_integ[valid, j, :] = nansum_pair(sz_integ[valid, j, :], w[:, None] * sz_psd)
                else:
                    sz_integ[valid, j, :] = nansum_pair(sz_integ[valid, j, :],
                                                        sz_psd * sb.quad_weight)

                if simulate_doppler and dop_scheme == 1:
                    vh, n = hyd[h].integrate_V()
                    v_integ[valid] = nansum_pair(v_integ[valid], vh)
                    n_integ[valid] = nansum_pair(n_integ[valid], n)
                elif simulate_doppler and dop_scheme == 3:
                    # attenuation of this hydrometeor at its valid gates (:297-305)
                    wl = K.Derived(config).WAVELENGTH
                    ah = 4.343e-3 * 2 * wl * sz_psd[:, 11]
                    ah *= radial_res / 1000.
                    ah_list.append(ah)
                elif simulate_doppler:
                    # scheme 2: fall speed weighted by N(D) x rcs_h, unit-spaced trapezoid
                    # (doppler_scatter.py:283-296)
                    rcs = 2 * np.pi * (sz[:, :, 0] - sz[:, :, 1] - sz[:, :, 2] + sz[:, :, 3])
                    v_f = hyd[h].get_V(list_D)
                    vh_w = np.trapezoid(np.multiply(v_f, N * rcs), axis=1)
                    n_w = np.trapezoid(N * rcs, axis=1)
                    v_integ[valid] = nansum_pair(v_integ[valid], vh_w)
                    n_integ[valid] = nansum_pair(n_integ[valid], n_w)

            if simulate_doppler and dop_scheme == 3:
                beam = SP.subbeam_spectrum(sb, hydrom_types, luts, config, varray)
                add = np.zeros(len(beam))
                if config['doppler']['turbulence_correction']:
                    raise NotImplementedError('oracle: turbulence correction needs EDR')
                if config['doppler']['motion_correction']:
                    add = add + SP.spectral_width_motion(sb.elev_profile, config)
                if np.sum(add) > 0:
                    beam = SP.broaden_spectrum(beam, add, varray)
                if mp['with_attenuation']:
                    beam = SP.apply_attenuation(beam, SP.attenuation_per_beam(ah_list, n_gates))
                # in place on the float32 spectrum, as the reference does (:386-389): the product
                # is rounded to float32 (and underflows there) before it is accumulated
                if not np.isscalar(sb.quad_weight):
                    beam *= sb.quad_weight[:, None]
                else:
                    beam *= sb.quad_weight
                doppler_spectrum += beam
            elif simulate_doppler:
                v_hydro = v_integ / n_integ
                theta = np.deg2rad(sb.elev_profile)
                phi = np.deg2rad(sb.quad_pt[0])
                proj = proj_vel(sb.values['U'], sb.values['V'], sb.values['W'], v_hydro, theta, phi)
                total_weight_rvel = total_weight_rvel + ~np.isnan(proj) * sb.quad_weight
                rvel_avg = nansum_pair(rvel_avg, proj * sb.quad_weight)

        sz_tot = np.nansum(sz_integ, axis=1)
        sz_tot[sz_tot == 0] = np.nan
        ZH, ZV, ZDR, RHOHV, KDP, AH, AV, DELTA_HV = pol_from_sz(sz_tot, config)
        PHIDP = nan_cumsum(2 * KDP) * radial_res / 1000. + DELTA_HV
        if mp['with_attenuation']:
            ZV_ATT = ZV.copy()
            ZH_ATT = ZH.copy()
            ZV_ATT *= nan_cumprod(10 ** (-0.1 * AV * (radial_res / 1000.)))
            ZH_ATT *= nan_cumprod(10 ** (-0.1 * AH * (radial_res / 1000.)))
            ZDR = ZH_ATT / ZV_ATT
        if simulate_doppler:
            if dop_scheme == 3:
                rvel_avg = SP.rvel_from_spectrum(doppler_spectrum, varray)
            else:
                rvel_avg /= total_weight_rvel
            if nyquist is not None:          # doppler_scatter.py:431-437
                rvel_avg = aliasing(rvel_avg, nyquist)

    obs = {'ZH': ZH, 'ZDR': ZDR, 'ZV': ZV, 'KDP': KDP, 'DELTA_HV': DELTA_HV, 'PHIDP': PHIDP,
           'RHOHV': RHOHV, 'ATT_H': AH, 'ATT_V': AV}
    if simulate_doppler:
        obs['RVEL'] = rvel_avg
        if dop_scheme == 3:
            obs['DSPECTRUM'] = doppler_spectrum

    mask = np.zeros(n_gates,)
    for sb in subbeams:
        mask = mask + sb.mask[0:n_gates]
    mask /= float(n_sub)
    mask[np.logical_and(mask > -1, mask <= 0)] = 0

    c = subbeams[idx_0]
    out = SubBeam(obs, mask, c.lats_profile, c.lons_profile, c.dist_profile, c.heights_profile)
    if return_sz:
        out.sz_integ = sz_integ
        out.sz_total = sz_tot
    return out


def sensitivity_threshold(config, n_gates):
    """dBZ threshold per gate, ranges = res*arange(n) (Q6), or None if the
    sensitivity spec is invalid (doppler_scatter.py:815-835)."""
    sens = config['radar']['sensitivity']
    if not isinstance(sens, list):
        sens = [sens]
    r = config['radar']['radial_resolution'] * np.arange(n_gates)
    with np.errstate(divide='ignore'):
        if len(sens) == 3:
            return (sens[0] + K.Derived(config).RADAR_CONSTANT_DB + sens[2]
                    + 20 * np.log10(r / 1000.))
        if len(sens) == 2:
            return (sens[0] - 20 * np.log10(sens[1] / 1000.)) + 20 * np.log10(r / 1000.)
        if len(sens) == 1:
            return sens[0] + 0 * r
    return None


def cut_at_sensitivity(radials, config):
    """In place (doppler_scatter.py:804-862).  Two branches, as in the reference:

    * a list of LISTS of radials -- what get_PPI / get_RHI pass (radar_operator.py:432-445,
      530-542): every simulated variable is censored with the gate mask
      10 log10(ZH) < threshold(r), EXCEPT the Doppler spectrum, which is censored bin by bin
      with 10 log10(DSPECTRUM) < threshold(r) (:839-850);
    * a simple list of radials (get_GPM_swath): everything, the spectrum rows included, with
      the gate mask (:852-861)."""
    if len(radials) and isinstance(radials[0], list):
        for sweep in radials:
            for b in sweep:
                thr = sensitivity_threshold(config, len(b.dist_profile))
                if thr is None:
                    return radials
                with np.errstate(invalid='ignore', divide='ignore'):
                    m = 10 * np.log10(b.values['ZH']) < thr
                    for k in b.values.keys():
                        if k not in K.SIMULATED_VARIABLES:
                            continue
                        if k == 'DSPECTRUM':
                            logspectrum = 10 * np.log10(b.values[k])
                            t2 = np.tile(thr, (logspectrum.shape[1], 1)).T
                            b.values[k][logspectrum < t2] = np.nan
                        else:
                            b.values[k][m] = np.nan
        return radials
    for b in radials:
        thr = sensitivity_threshold(config, len(b.dist_profile))
        if thr is None:
            return radials
        with np.errstate(invalid='ignore', divide='ignore'):
            m = 10 * np.log10(b.values['ZH']) < thr
        for k in b.values.keys():
            if k in K.SIMULATED_VARIABLES:
                b.values[k][m] = np.nan
    return radials
