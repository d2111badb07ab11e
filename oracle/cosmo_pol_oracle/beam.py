"""Oracle beam geometry + gate interpolation (TEST INFRASTRUCTURE ONLY).

Restates, per radial, the reference's
  get_earth_radius                cosmo_pol/utilities/utilities.py:126-139
  _ref_4_3                        cosmo_pol/interpolation/atm_refraction.py:181-220
  quadrature scheme 1 + threshold cosmo_pol/interpolation/interpolation.py:148-166,342-354
  sub-beam loop / mask coding     cosmo_pol/interpolation/interpolation.py:361-438
  trilin_interp_radial            cosmo_pol/interpolation/interpolation.py:498-597
  melting                         cosmo_pol/interpolation/melting.py:19-90
  integrate_radials               cosmo_pol/interpolation/interpolation.py:36-89
The C gate kernel is called through ctypes (oracle/interp_twin.c, or the
compiled reference in oracle/_ref when asked).
"""
import ctypes
import os

import numpy as np

from . import constants as K
from . import geodesy

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIBS = {}


def _load_interp_lib(which='twin'):
    if which in _LIBS:
        return _LIBS[which]
    base = os.path.dirname(_HERE)
    path = (os.path.join(base, '_build', 'libinterp_twin.so') if which == 'twin'
            else os.path.join(base, '_ref', 'libinterp_ref.so'))
    if not os.path.exists(path):
        import subprocess
        subprocess.check_call(['make', '-s', '-C', base, 'twin' if which == 'twin' else 'ref'])
    lib = ctypes.CDLL(path)
    fp = ctypes.POINTER(ctypes.c_float)
    lib.get_all_radar_pts.restype = fp
    lib.get_all_radar_pts.argtypes = [fp, ctypes.c_int, fp, ctypes.c_int, ctypes.c_int,
                                      fp, ctypes.c_int, fp, ctypes.c_int, ctypes.c_int,
                                      ctypes.c_int, fp, ctypes.c_int, ctypes.c_int,
                                      ctypes.c_int, fp, ctypes.c_int, fp, ctypes.c_int]
    _LIBS[which] = lib
    return lib


def get_all_radar_pts(coords, heights, data, zlevels, llc, res, which='twin'):
    """ctypes call with the reference's C signature (interpolation_c.c:8)."""
    lib = _load_interp_lib(which)
    fp = ctypes.POINTER(ctypes.c_float)
    c32 = lambda a: np.ascontiguousarray(a, dtype=np.float32)
    coords, heights, data, zlevels, llc, res = map(c32, (coords, heights, data, zlevels, llc, res))
    n = coords.shape[0]
    out = np.empty(n, dtype=np.float32)
    P = lambda a: a.ctypes.data_as(fp)
    lib.get_all_radar_pts(P(out), n, P(coords), n, coords.shape[1], P(heights), heights.shape[0],
                          P(data), data.shape[0], data.shape[1], data.shape[2],
                          P(zlevels), zlevels.shape[0], zlevels.shape[1], zlevels.shape[2],
                          P(llc), 2, P(res), 2)
    return out


def earth_radius(latitude):
    # NB quirk Q1: the argument is used as if it were radians (utilities.py:135-138)
    a = 6378.1370 * 1000
    b = 6356.7523 * 1000
    num = ((a ** 2 * np.cos(latitude)) ** 2 + (b ** 2 * np.sin(latitude)) ** 2)
    den = ((a * np.cos(latitude)) ** 2 + (b * np.sin(latitude)) ** 2)
    return np.sqrt(num / den)


def trajectory_4_3(range_vec, elevation_deg, coords_radar):
    """(s, h, e) of the 4/3-earth model; quirk Q1: the *longitude*
    coords_radar[1] feeds the earth radius (atm_refraction.py:201)."""
    el = np.deg2rad(elevation_deg)
    alt = coords_radar[2]
    RE = earth_radius(coords_radar[1])
    ke_re = K.KE * RE
    temp = np.sqrt(range_vec ** 2 + ke_re ** 2 + 2 * range_vec * K.KE * RE * np.sin(el))
    h = temp - K.KE * RE + alt
    s = K.KE * RE * np.arcsin((range_vec * np.cos(el)) / (K.KE * RE + h))
    e = el + np.arctan(range_vec * np.cos(el) / (range_vec * np.sin(el) + K.KE * RE + alt))
    return s.astype('float32'), h.astype('float32'), np.rad2deg(e.astype('float32'))


def gauss_hermite_subbeams(config):
    """Scheme-1 antenna quadrature: returns (pts_hor, pts_ver, weights[nh,nv],
    keep[nh,nv]) (interpolation.py:148-166, 342-354, 383)."""
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
    broadening = nh > 1 or nv > 1
    w_sorted = np.sort(np.array(weights).ravel())[::-1]
    w_cum = np.cumsum(w_sorted / np.sum(w_sorted))
    w_cum[-1] = 1.
    idx_above = np.where(w_cum >= config['integration']['weight_threshold'])[0][0]
    threshold = w_sorted[idx_above]
    keep = (weights >= threshold) | (not broadening)
    return pts_hor, pts_ver, weights, keep


class SubBeam(object):
    """One antenna-quadrature point of one radial (reference: Radial,
    cosmo_pol/interpolation/radial.py:16-54)."""

    def __init__(self, values, mask, lats, lons, dist, heights, elev=None, quad_pt=None,
                 quad_weight=1):
        self.values = values
        self.mask = mask
        self.lats_profile = lats
        self.lons_profile = lons
        self.dist_profile = dist
        self.heights_profile = heights
        self.elev_profile = elev
        self.quad_pt = quad_pt
        self.quad_weight = quad_weight
        self.has_melting = False
        self.mask_ml = None


class ModelCube(object):
    """Model variables on the rotated grid: data[name] [nz,ny,nx] f32,
    zlevels [nzl,ny,nx] f32 (index 0 = model top), proj_info dict
    (Lo1, La1, Lo2, La2, Latitude/Longitude_of_southern_pole), resolution
    (dlon, dlat). Duck-types what interpolation.py:547-561 reads."""

    def __init__(self, data, zlevels, proj_info, resolution, order):
        self.data = data
        self.zlevels = zlevels
        self.proj_info = proj_info
        self.resolution = np.asarray(resolution, dtype=np.float32)
        self.order = list(order)

    @property
    def llc(self):
        return np.asarray((float(self.proj_info['Lo1']), float(self.proj_info['La1']))).astype('float32')

    @property
    def urc(self):
        return np.asarray((float(self.proj_info['Lo2']), float(self.proj_info['La2']))).astype('float32')


def gate_coordinates(cube, coords_radar, azimuth, dist):
    """lat/lon (float64) and rotated coords (float32 [n,2]) of every gate
    (interpolation.py:526-568); raises IndexError outside the model domain
    (interpolation.py:572-580)."""
    lats, lons = geodesy.wgs84_direct(coords_radar[0], coords_radar[1], azimuth,
                                      np.asarray(dist, dtype=np.float64))
    p = cube.proj_info
    rc = geodesy.wgs_to_rotated(lats, lons, p['Latitude_of_southern_pole'],
                                p['Longitude_of_southern_pole'])
    llc, urc = cube.llc, cube.urc
    if (np.any(rc[:, 1] < llc[0]) or np.any(rc[:, 0] < llc[1]) or
            np.any(rc[:, 1] > urc[0]) or np.any(rc[:, 0] > urc[1])):
        raise IndexError('RADAR DOMAIN IS NOT ENTIRELY CONTAINED IN COSMO SIMULATION DOMAIN')
    return lats, lons, rc


def _one_subbeam(cube, config, coords_radar, az_pt, el_pt, traj, weight, which):
    s, h, e = traj
    lats, lons, rc = gate_coordinates(cube, coords_radar, az_pt, s)
    vals = {}
    mask = None
    for k, name in enumerate(cube.order):
        b = get_all_radar_pts(rc, h, cube.data[name], cube.zlevels, cube.llc,
                              cube.resolution, which=which)
        if k == 0:       # mask from the FIRST variable only (:401-409)
            mask = np.zeros((len(b)))
            mask[b == -9999] = 1
            mask[np.isnan(b)] = -1
        b[mask != 0] = np.nan
        vals[name] = b
    # NB the reference shares the trajectory arrays between sub-beams with
    # the same vertical node; the in-place elevation fold (quirk Q8) is
    # idempotent, so private copies give the same numbers.
    sb = SubBeam(vals, mask, lats, lons, s, h, e.copy(), [az_pt, el_pt], weight)
    if config['microphysics']['with_melting']:
        apply_melting(sb)
    return sb


def interpolate_radial(cube, config, azimuth, elevation, which='twin', trajs=None,
                       coords_radar=None):
    """All kept sub-beams of one radial (regular quadratures: horizontal index outer,
    vertical inner; schemes with point lists: list order), each with mask coding and
    optional melting.  `trajs`: optional list of (s, h, e) per vertical node (per
    point for list quadratures) replacing the 4/3-earth model (spaceborne rays, ODE
    refraction); `coords_radar`: site of this ray."""
    from .quadrature import ml_edge_mask, quadrature
    q = quadrature(config)
    if coords_radar is None:
        coords_radar = config['radar']['coords']
    if trajs is None:
        der = K.Derived(config)
        trajs = [trajectory_4_3(der.RANGE_RADAR, pt + elevation, coords_radar) for pt in q.pts_ver]
    out = []
    if q.kind == 'list':
        for i in range(len(q.weights)):
            if q.weights[i] >= q.threshold:
                out.append(_one_subbeam(cube, config, coords_radar, q.pts_hor[i] + azimuth,
                                        q.pts_ver[i] + elevation, trajs[i], q.weights[i], which))
        return out
    for i in range(len(q.pts_hor)):
        for j in range(len(q.pts_ver)):
            if not (q.weights[i, j] >= q.threshold or not q.broadening):
                continue
            sb = _one_subbeam(cube, config, coords_radar, q.pts_hor[i] + azimuth,
                              q.pts_ver[j] + elevation, trajs[j], q.weights[i, j], which)
            if q.ml_nv is not None:          # scheme 'ml' (interpolation.py:423-436)
                n = len(sb.lats_profile)
                if j > q.ml_nv:
                    sb.quad_weight = sb.quad_weight * ml_edge_mask(sb.mask_ml, n)
                else:
                    sb.quad_weight = sb.quad_weight * np.ones(n)
            out.append(sb)
    return out


def apply_melting(sb):
    """melting.py:19-90 (in place)."""
    v = sb.values
    T, QR, QS, QG = v['T'], v['QR_v'], v['QS_v'], v['QG_v']
    n = len(T)
    with np.errstate(invalid='ignore', divide='ignore'):
        ml = np.logical_and(QR > 0, (QS + QG) > 0)
        for name in ('QmS_v', 'QmG_v', 'fwet_mS', 'fwet_mG'):
            v[name] = np.zeros((n))
        if np.any(ml):
            qr, qs, qg = QR[ml], QS[ml], QG[ml]
            qsg = qs + qg
            v['QmS_v'][ml] = qs + qr * (qs / qsg)
            v['QmG_v'][ml] = qg + qr * (qg / qsg)
            wet = np.logical_or(v['QmS_v'] > 0, v['QmG_v'] > 0)
            QS[wet] = 0
            QG[wet] = 0
            QR[wet] = 0
            v['fwet_mS'][ml] = (qr * qs / qsg) / v['QmS_v'][ml]
            v['fwet_mG'][ml] = (qr * qg / qsg) / v['QmG_v'][ml]
            sb.has_melting = True
        else:
            sb.has_melting = False
    sb.mask_ml = ml
    return sb


def nansum_pair(x, y):
    """utilities.py:231-261 for equal shapes / length-1 seeds."""
    x = np.array(x)
    y = np.array(y)
    if x.shape != y.shape:
        pad = [(0, max(0, d2 - d1)) for d1, d2 in zip(x.shape, y.shape)]
        x = np.pad(x, pad, 'constant', constant_values=0)
        pad = [(0, max(0, d1 - d2)) for d1, d2 in zip(x.shape, y.shape)]
        y = np.pad(y, pad, 'constant', constant_values=0)
    return np.nansum([x, y], axis=0)


def integrate_subbeams(subbeams):
    """Model variables averaged over the antenna pattern
    (interpolation.py:36-89)."""
    n = len(subbeams)
    sum_w = 0
    for sb in subbeams:
        sum_w += sb.quad_weight
    out = {}
    for k in subbeams[0].values.keys():
        acc = np.array([np.nan])
        for sb in subbeams:
            acc = nansum_pair(acc, sb.values[k] * sb.quad_weight / sum_w)
        out[k] = acc
    mask = np.zeros(len(subbeams[0].mask))
    for sb in subbeams:
        mask = mask + sb.mask
    mask /= float(n)
    mask[np.logical_and(mask > -1, mask <= 0)] = 0
    c = subbeams[int(n / 2.)]
    return SubBeam(out, mask, c.lats_profile, c.lons_profile, c.dist_profile, c.heights_profile)
