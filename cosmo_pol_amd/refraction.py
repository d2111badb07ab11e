"""Refraction scheme 2: ray paths from the Zeng & Blahak (2014) ODE.

Host-side, one solve per distinct (elevation + vertical quadrature node), exactly
as the reference does it -- scipy's LSODA over scipy's interp1d of the refractivity column,
on the dtypes the model fields come in (cosmo_pol/interpolation/atm_refraction.py:56-179):
the resulting float32 (s, h, e) tables are handed to the GPU through CPOL_GEOM_HOST_PATHS.  Quirks kept: the refractivity column is
read at [round(p0), round(p0)] (the row index is used for both axes, :113-117)
and the earth radius is evaluated with the radar LATITUDE in degrees fed to
cos/sin (:120, utilities.py:126-139).
"""
import numpy as np
from scipy.integrate import odeint

from .geometry import DEG, get_earth_radius


def wgs_to_rotated(lat_deg, lon_deg, sp_lat_deg, sp_lon_deg):
    """Geographic -> rotated-pole (rlat, rlon) in degrees (float64); stands in
    for pycosmo.WGS_to_COSMO (call site atm_refraction.py:101-103)."""
    theta = (90.0 + np.float64(sp_lat_deg)) * DEG
    phi = np.float64(sp_lon_deg) * DEG
    ct, st, cp, sp = np.cos(theta), np.sin(theta), np.cos(phi), np.sin(phi)
    lat = np.asarray(lat_deg, dtype=np.float64) * DEG
    lon = np.asarray(lon_deg, dtype=np.float64) * DEG
    cl = np.cos(lat)
    x, y, z = np.cos(lon) * cl, np.sin(lon) * cl, np.sin(lat)
    x_new = ct * cp * x + ct * sp * y + st * z
    y_new = -sp * x + cp * y
    z_new = -st * cp * x - st * sp * y + ct * z
    return np.arcsin(z_new) / DEG, np.arctan2(y_new, x_new) / DEG


class _PiecewiseLinear(object):
    """Linear interpolation with linear extrapolation beyond both ends
    (atm_refraction.py:151-179): scipy's interp1d inside the table, as the reference builds it --
    LSODA's step control amplifies a last-bit difference of the right-hand side into float32-ulp
    differences of the path, so the interpolant is the reference's own, not a re-derivation."""

    def __init__(self, x, y):
        from scipy.interpolate import interp1d
        self.f = interp1d(x, y)                 # (dtypes as given: a float32 column interpolates with float32 slopes)
        self.x, self.y = self.f.x, self.f.y

    def __call__(self, v):
        x, y = self.x, self.y
        if v < x[0]:
            return y[0] + (v - x[0]) * (y[1] - y[0]) / (x[1] - x[0])
        if v > x[-1]:
            return y[-1] + (v - x[-1]) * (y[-1] - y[-2]) / (x[-1] - x[-2])
        return float(self.f(v))


def refractivity_column(N_data, zlevels, proj_info, resolution, coords_radar, radar_type='ground'):
    rlat, rlon = wgs_to_rotated(coords_radar[0], coords_radar[1],
                                proj_info['Latitude_of_southern_pole'],
                                proj_info['Longitude_of_southern_pole'])
    rlat, rlon = np.float32(rlat), np.float32(rlon)      # WGS_to_COSMO returns float32
    llc = (float(proj_info['Lo1']), float(proj_info['La1']))
    p0 = (rlat - llc[1]) / resolution[1]
    i = int(np.round(p0))
    n_prof = 1 + (N_data[:, i, i]) * 1E-6
    h = zlevels[:, i, i]
    if radar_type == 'ground':
        h, n_prof = h[::-1], n_prof[::-1]
    # (dtypes as the reference leaves them: float32 model fields give a float32 column, and the slopes
    # diff(n) / diff(h) of ode_path are then float32 quotients, atm_refraction.py:122-123)
    return np.asarray(h), np.asarray(n_prof)


def ode_path(range_vec, elevation_deg, coords_radar, h_col, n_col):
    """(s, h, e_deg) float32 along one ray (atm_refraction.py:119-148)."""
    RE = get_earth_radius(coords_radar[0])
    n_of_h = _PiecewiseLinear(h_col, n_col)
    dn_dh = _PiecewiseLinear(h_col[0:-1], np.diff(n_col) / np.diff(h_col))

    def deriv(z, r):
        hh, u = z
        n = n_of_h(hh)
        dn = dn_dh(hh)
        return [u, (-u ** 2 * ((1. / n) * dn + 1. / (RE + hh)) + ((1. / n) * dn + 1. / (RE + hh)))]
    z0 = [coords_radar[2], np.sin(np.deg2rad(elevation_deg))]
    Z = odeint(deriv, z0, range_vec)
    h = Z[:, 0]
    e = np.arcsin(Z[:, 1])
    s = np.zeros(h.shape)
    dR = range_vec[1] - range_vec[0]
    for i in range(1, len(s)):
        s[i] = s[i - 1] + RE * np.arcsin((np.cos(e[i - 1]) * dR) / (RE + h[i]))
    return s.astype('float32'), h.astype('float32'), np.rad2deg(e.astype('float32'))


def ode_paths(range_vec, elevations, pts_ver, coords_radar, h_col, n_col):
    """float32 [n_rays, n_vnodes, 3, n_gates]; solves are shared between rays with
    equal elevation (a PPI needs n_vnodes solves)."""
    el = np.asarray(elevations, dtype=np.float64)
    out = np.zeros((len(el), len(pts_ver), 3, len(range_vec)), dtype=np.float32)
    cache = {}
    for r in range(len(el)):
        for j, pt in enumerate(pts_ver):
            key = float(pt + el[r])
            if key not in cache:
                cache[key] = np.stack(ode_path(range_vec, key, coords_radar, h_col, n_col))
            out[r, j] = cache[key]
    return out
