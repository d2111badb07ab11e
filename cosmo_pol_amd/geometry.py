"""Host-side (per-ray, not per-gate) geometry of a sweep.

Everything that depends on the gate index runs on the GPU; this module only
evaluates, with NumPy float64, the handful of per-ray / per-quadrature-node
constants the kernels consume, in the same operation order as the reference so
that the float32 ray paths match bit for bit:

  quadrature scheme 1 (Gauss-Hermite)  cosmo_pol/interpolation/interpolation.py:148-166
  weight threshold / kept sub-beams    cosmo_pol/interpolation/interpolation.py:342-354, 383
  earth radius (quirk Q1)              cosmo_pol/utilities/utilities.py:126-139,
                                       cosmo_pol/interpolation/atm_refraction.py:201-204
  deg2rad / sin / cos of the elevation cosmo_pol/interpolation/atm_refraction.py:197, 206-215
  Vincenty direct, per-ray constants   stands in for pyproj.Geod.fwd,
                                       cosmo_pol/interpolation/interpolation.py:526-536
"""
import numpy as np

from . import constants as K

WGS84_A = 6378137.0
WGS84_F = 1.0 / 298.257223563
WGS84_B = (1.0 - WGS84_F) * WGS84_A
DEG = np.pi / 180.0


def get_earth_radius(latitude):
    """WGS84 radius formula; NB the reference feeds the argument (in degrees)
    straight to cos/sin and passes the radar LONGITUDE (quirk Q1) - kept."""
    a = 6378.1370 * 1000
    b = 6356.7523 * 1000
    num = ((a ** 2 * np.cos(latitude)) ** 2 + (b ** 2 * np.sin(latitude)) ** 2)
    den = ((a * np.cos(latitude)) ** 2 + (b * np.sin(latitude)) ** 2)
    return np.sqrt(num / den)


class SubBeams(object):
    """Antenna quadrature of a radial: kept (horizontal node, vertical node,
    weight) triples in the reference's loop order (horizontal outer)."""

    def __init__(self, pts_hor, pts_ver, weights, keep):
        self.pts_hor = np.asarray(pts_hor, dtype=np.float64)
        self.pts_ver = np.asarray(pts_ver, dtype=np.float64)
        self.weights = weights
        ih, jv = np.nonzero(keep)
        self.sub_h = ih.astype(np.int32)
        self.sub_v = jv.astype(np.int32)
        self.sub_w = np.ascontiguousarray(weights[ih, jv], dtype=np.float64)
        # scheme 'ml' only: sub-beams whose weight is w x (smoothed melting-layer edge mask)
        self.sub_smooth = None
        self.ml_radius, self.ml_filter = 0, None

    @classmethod
    def from_points(cls, off_hor, off_ver, weights):
        """Irregular quadratures (point lists): every sub-beam has its own azimuth and
        elevation offset, i.e. horizontal node k pairs with vertical node k only."""
        n = len(weights)
        sb = cls(off_hor, off_ver, np.diag(np.asarray(weights, dtype=np.float64)), np.eye(n, dtype=bool))
        return sb

    @property
    def n_sub(self):
        return len(self.sub_w)

    @property
    def central(self):
        return int(self.n_sub / 2)


def gauss_hermite_subbeams(config):
    """Scheme 1 only (kept for callers that want the plain Gauss-Hermite grid);
    `quadrature.subbeams` dispatches on integration/scheme."""
    bw = config['radar']['3dB_beamwidth']
    nh = int(config['integration']['nh_GH'])
    nv = int(config['integration']['nv_GH'])
    sigma = bw / (2 * np.sqrt(2 * np.log(2)))
    pts_hor, w_hor = np.polynomial.hermite.hermgauss(nh)
    pts_hor = pts_hor * sigma
    pts_ver, w_ver = np.polynomial.hermite.hermgauss(nv)
    pts_ver = pts_ver * sigma
    weights = np.outer(w_hor * sigma, w_ver * sigma)
    weights *= np.abs(np.cos(np.deg2rad(pts_ver)))
    weights /= np.sum(weights.ravel())
    # threshold on the cumulated sorted weights; weights are NOT renormalised
    w_sorted = np.sort(np.array(weights).ravel())[::-1]
    w_cum = np.cumsum(w_sorted / np.sum(w_sorted))
    w_cum[-1] = 1.
    threshold = w_sorted[np.where(w_cum >= config['integration']['weight_threshold'])[0][0]]
    keep = weights >= threshold
    if not (nh > 1 or nv > 1):
        keep[:] = True
    return SubBeams(pts_hor, pts_ver, weights, keep)


def radar_site_constants(coords):
    """Reduced latitude of the radar (sin U1, cos U1, tan U1) for Vincenty direct;
    `coords` = [lat, lon, alt] or an array [n_rays, 3] (one site per ray)."""
    lat = np.asarray(coords, dtype=np.float64)[..., 0]
    phi1 = lat * DEG
    tan_u1 = (1.0 - WGS84_F) * np.tan(phi1)
    cos_u1 = 1.0 / np.sqrt(1.0 + tan_u1 * tan_u1)
    sin_u1 = tan_u1 * cos_u1
    if np.ndim(sin_u1) == 0:
        return float(sin_u1), float(cos_u1), float(tan_u1)
    return sin_u1, cos_u1, tan_u1


def ray_tables(coords, azimuths, elevations, sub):
    """traj [n_rays, n_v, 4] = (el_rad, sin el, cos el, el_deg) and geo [n_rays, n_h, 8]
    = (sin a1, cos a1, sigma1, sin alpha, b*A, B, C, a1) for every ray.  `coords` is the
    radar site [lat, lon, alt] or one site per ray [n_rays, 3] (spaceborne)."""
    az = np.asarray(azimuths, dtype=np.float64).reshape(-1)
    el = np.asarray(elevations, dtype=np.float64).reshape(-1)
    n = len(az)
    # reference: compute_trajectory_radial(rranges, pt + elevation, ...)
    el_deg = sub.pts_ver[None, :] + el[:, None]
    el_nodes = np.deg2rad(el_deg)
    traj = np.stack([el_nodes, np.sin(el_nodes), np.cos(el_nodes), el_deg], axis=-1)
    sin_u1, cos_u1, tan_u1 = radar_site_constants(coords)
    if np.ndim(sin_u1) == 1:
        cos_u1, tan_u1 = cos_u1[:, None], tan_u1[:, None]
    a, b, f = WGS84_A, WGS84_B, WGS84_F
    alpha1 = (sub.pts_hor[None, :] + az[:, None]) * DEG
    sin_a1 = np.sin(alpha1)
    cos_a1 = np.cos(alpha1)
    sigma1 = np.arctan2(tan_u1, cos_a1)
    sin_alpha = cos_u1 * sin_a1
    cos2_alpha = 1.0 - sin_alpha * sin_alpha
    u2 = cos2_alpha * (a * a - b * b) / (b * b)
    A = 1.0 + u2 / 16384.0 * (4096.0 + u2 * (-768.0 + u2 * (320.0 - 175.0 * u2)))
    B = u2 / 1024.0 * (256.0 + u2 * (-128.0 + u2 * (74.0 - 47.0 * u2)))
    C = f / 16.0 * cos2_alpha * (4.0 + f * (4.0 - 3.0 * cos2_alpha))
    geo = np.stack([sin_a1, cos_a1, sigma1, sin_alpha, b * A, B, C, alpha1], axis=-1)
    assert traj.shape == (n, len(sub.pts_ver), 4) and geo.shape == (n, len(sub.pts_hor), 8)
    return np.ascontiguousarray(traj), np.ascontiguousarray(geo)


def sensitivity_threshold(config, derived, n_gates):
    """dBZ censoring threshold per gate (cut_at_sensitivity,
    cosmo_pol/scatter/doppler_scatter.py:815-842; ranges start at 0, quirk Q6).
    Returns None when the specification is invalid (reference: prints, no cut)."""
    sens = config['radar']['sensitivity']
    if not isinstance(sens, list):
        sens = [sens]
    r = config['radar']['radial_resolution'] * np.arange(n_gates)
    with np.errstate(divide='ignore'):
        if len(sens) == 3:
            thr = sens[0] + derived.RADAR_CONSTANT_DB + sens[2] + 20 * np.log10(r / 1000.)
        elif len(sens) == 2:
            thr = (sens[0] - 20 * np.log10(sens[1] / 1000.)) + 20 * np.log10(r / 1000.)
        elif len(sens) == 1:
            thr = sens[0] + 0.0 * r
        else:
            print('Sensitivity parameters are invalid, cannot cut at specified sensitivity')
            return None
    return np.ascontiguousarray(thr, dtype=np.float64)


def earth_radius_for_refraction(coords):
    return float(get_earth_radius(coords[1])), float(K.KE)
