"""Oracle restatement of refraction scheme 2 (TEST INFRASTRUCTURE ONLY):
cosmo_pol/interpolation/atm_refraction.py:56-179 (_deriv_z, _ref_ODE,
_piecewise_linear), scipy odeint + interp1d as upstream, including the column
index quirk (:113-117) and the latitude-in-degrees earth radius (:120)."""
import numpy as np
from scipy.integrate import odeint
from scipy.interpolate import interp1d

from . import beam, geodesy


def _piecewise_linear(x, y):
    f = interp1d(x, y)
    xs, ys = f.x, f.y

    def point(v):
        if v < xs[0]:
            return ys[0] + (v - xs[0]) * (ys[1] - ys[0]) / (xs[1] - xs[0])
        if v > xs[-1]:
            return ys[-1] + (v - xs[-1]) * (ys[-1] - ys[-2]) / (xs[-1] - xs[-2])
        return f(v)

    def many(v):
        if np.isscalar(v):
            v = [v]
        return np.array([point(q) for q in v])
    return many


def trajectory_ode(range_vec, elevation_deg, coords_radar, N_data, zlevels, proj_info, resolution,
                   radar_type='ground'):
    rc = geodesy.wgs_to_rotated(np.array([coords_radar[0]]), np.array([coords_radar[1]]),
                                proj_info['Latitude_of_southern_pole'],
                                proj_info['Longitude_of_southern_pole'])[0]
    llc = (float(proj_info['Lo1']), float(proj_info['La1']))
    pos = [(rc[0] - llc[1]) / resolution[1], (rc[1] - llc[0]) / resolution[0]]
    i = int(np.round(pos[0]))
    n_prof = 1 + (N_data[:, i, i]) * 1E-6
    h = zlevels[:, i, i]
    RE = beam.earth_radius(coords_radar[0])
    if radar_type == 'ground':
        h = h[::-1]
        n_prof = n_prof[::-1]
    n_h = _piecewise_linear(h, n_prof)
    dn_dh = _piecewise_linear(h[0:-1], np.diff(n_prof) / np.diff(h))

    def deriv(z, r):
        hh, u = z
        n = float(n_h(hh)[0])       # upstream keeps 1-element arrays (ragged -> odeint raises)
        d = float(dn_dh(hh)[0])
        return [u, (-u ** 2 * ((1. / n) * d + 1. / (RE + hh)) + ((1. / n) * d + 1. / (RE + hh)))]
    Z = odeint(deriv, [coords_radar[2], np.sin(np.deg2rad(elevation_deg))], range_vec)
    hh = Z[:, 0]
    e = np.arcsin(Z[:, 1])
    s = np.zeros(hh.shape)
    dR = range_vec[1] - range_vec[0]
    for k in range(1, len(s)):
        s[k] = s[k - 1] + RE * np.arcsin((np.cos(e[k - 1]) * dR) / (RE + hh[k]))
    return s.astype('float32'), hh.astype('float32'), np.rad2deg(e.astype('float32'))
