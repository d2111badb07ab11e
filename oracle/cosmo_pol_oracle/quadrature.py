"""Oracle antenna quadratures (TEST INFRASTRUCTURE ONLY).

Restates the weight / node part of get_interpolated_radial
(cosmo_pol/interpolation/interpolation.py):
  scheme 1   Gauss-Hermite                         :148-166
  scheme 'ml' Gauss-Hermite, 10x vertical nodes sorted by weight  :168-193
  scheme 2   multi-Gaussian antenna fit, polar Gauss-Hermite x Gauss-Legendre  :195-229
  scheme 3   Gauss-Legendre weighted by a measured antenna diagram  :231-256
  threshold on the cumulated sorted weights        :342-354
and utilities.py:285-341 (polar2cartesian / vector_1d_to_polar),
interpolation/antenna_fit.py:74-112 (optimize_gaussians; the reference's
objective reshapes with a float under Python 3 and raises -- restated with the
integer division it intends, parity unpinned).
Schemes 4 (hard-coded pickle of the authors' disk), 5 and 6 (rejected by the
reference's own VALID_VALUES, cfg.py:104) are not restated.
"""
import numpy as np

SQRT_8LN2 = 2 * np.sqrt(2 * np.log(2))


class Quadrature(object):
    """kind 'grid': weights[nh, nv] over pts_hor x pts_ver (offsets, deg);
    kind 'list': weights[n] with offsets pts_hor[n], pts_ver[n]."""

    def __init__(self, kind, pts_hor, pts_ver, weights, threshold, broadening=True, ml_nv=None):
        self.kind = kind
        self.pts_hor = np.asarray(pts_hor, dtype=np.float64)
        self.pts_ver = np.asarray(pts_ver, dtype=np.float64)
        self.weights = weights
        self.threshold = threshold
        self.broadening = broadening
        self.ml_nv = ml_nv          # 'ml': vertical nodes with index > ml_nv get edge-mask weights


def _threshold(weights, config):
    w_sorted = np.sort(np.array(weights).ravel())[::-1]
    w_cum = np.cumsum(w_sorted / np.sum(w_sorted))
    w_cum[-1] = 1.
    idx_above = np.where(w_cum >= config['integration']['weight_threshold'])[0][0]
    return w_sorted[idx_above]


def polar2cartesian(r, t, grid, x, y, order=3):
    from scipy.interpolate import interp1d
    from scipy.ndimage import map_coordinates
    X, Y = np.meshgrid(x, y)
    new_r = np.sqrt(X * X + Y * Y)
    new_t = np.arctan2(X, Y) + np.pi
    ir = interp1d(r, np.arange(len(r)), bounds_error=False)
    it = interp1d(t, np.arange(len(t)))
    new_ir = ir(new_r.ravel())
    new_it = it(new_t.ravel())
    new_ir[new_r.ravel() > r.max()] = len(r) - 1
    new_ir[new_r.ravel() < r.min()] = 0
    return map_coordinates(grid, np.array([new_it, new_ir]), order=order).reshape(new_r.shape)


def vector_1d_to_polar(angles, values, x, y):
    midpt = int(np.floor(len(angles) / 2.))
    r = angles[midpt:]
    thet = [0, np.pi, 2 * np.pi]
    pol = np.zeros((len(thet), len(r)))
    pol[0, :] = values[midpt:]
    pol[1, :] = values[0:midpt + 1]
    pol[1, :] = pol[1, ::-1]
    pol[2, :] = pol[0, :]
    return polar2cartesian(r, thet, pol, x, y)


def optimize_gaussians(x, y, n_gaussians):
    from scipy.optimize import minimize
    from scipy.signal import argrelextrema

    def gaussian_sum(x, params):
        return 10 * np.log10(np.sum([10 ** (0.1 * p[0]) * np.exp(-(x - p[1]) ** 2 / (2 * p[2] ** 2))
                                     for p in params], axis=0))

    def obj(params, x, y):
        params = np.reshape(params, (len(params) // 3, 3))
        return np.sqrt(np.sum((gaussian_sum(x, params) - y) ** 2))

    peaks = argrelextrema(y, np.greater)
    a_lobes, mu_lobes = y[peaks], x[peaks]
    if 0 not in mu_lobes:
        mu_lobes = np.append(mu_lobes, 0)
        a_lobes = np.append(a_lobes, 0)
    params = np.column_stack((a_lobes, mu_lobes))
    params = np.flipud(params[params[:, 0].argsort()])
    selected = params[0:n_gaussians, :]
    p0 = np.column_stack((selected[:, 0], selected[:, 1], np.array([0.5] * n_gaussians)))
    bounds = []
    for i in range(n_gaussians):
        for j in range(3):
            bounds.append([None, None] if j != 2 else [0.1, 2])
    bounds[0] = [0, 0]
    bounds[1] = [0, 0]
    res = minimize(obj, p0.ravel(), args=(x, y), bounds=bounds, method='SLSQP')
    return np.reshape(res['x'], (n_gaussians, 3))


def quadrature(config):
    integ = config['integration']
    scheme = integ['scheme']
    bw = config['radar']['3dB_beamwidth']
    if scheme in (1, 'ml'):
        nh, nv = int(integ['nh_GH']), int(integ['nv_GH'])
        nv_nodes = nv
        if scheme == 'ml':
            nv_nodes = 10 * nv
            if not nv_nodes % 2:
                nv_nodes += 1
        sigma = bw / SQRT_8LN2
        pts_hor, w_hor = np.polynomial.hermite.hermgauss(nh)
        pts_hor = pts_hor * sigma
        pts_ver, w_ver = np.polynomial.hermite.hermgauss(nv_nodes)
        pts_ver = pts_ver * sigma
        if scheme == 'ml':
            idx_sort = np.argsort(w_ver)[::-1]
            w_ver = w_ver[idx_sort]
            pts_ver = pts_ver[idx_sort]
        weights = np.outer(w_hor * sigma, w_ver * sigma)
        weights *= np.abs(np.cos(np.deg2rad(pts_ver)))
        weights /= np.sum(weights.ravel())
        return Quadrature('grid', pts_hor, pts_ver, weights, _threshold(weights, config),
                          nh > 1 or nv > 1, nv if scheme == 'ml' else None)
    if scheme == 2:
        nr, na = int(integ['nr_GH']), int(integ['na_GL'])
        ap = np.asarray(integ['antenna_params'], dtype=np.float64)
        pts_ang, w_ang = np.polynomial.legendre.leggauss(na)
        pts_rad, w_rad = np.polynomial.hermite.hermgauss(nr)
        a_dB, mu, sigma = ap[:, 0], ap[:, 1], ap[:, 2]
        ph, pv, weights = [], [], []
        sum_weights = 0
        for i in range(nr):
            for j in range(len(sigma)):
                for k in range(na):
                    r = mu[j] + np.sqrt(2) * sigma[j] * pts_rad[i]
                    theta = np.pi * pts_ang[k] + np.pi
                    weight = (np.pi * w_ang[k] * w_rad[i] * 10 ** (0.1 * a_dB[j])
                              * np.sqrt(2) * sigma[j] * abs(r))
                    weight *= np.cos(r * np.sin(theta))
                    weights.append(weight)
                    sum_weights += weight
                    ph.append(r * np.cos(theta))
                    pv.append(r * np.sin(theta))
        weights = np.array(weights)
        weights /= sum_weights
        return Quadrature('list', ph, pv, weights, _threshold(weights, config))
    if scheme == 3:
        nh, nv = int(integ['nh_GH']), int(integ['nv_GH'])
        antenna = np.genfromtxt(integ['antenna_diagram'], delimiter=',')
        angles = antenna[:, 0]
        power_sq = (10 ** (0.1 * antenna[:, 1])) ** 2
        bounds = np.max(angles)
        pts_hor, w_hor = np.polynomial.legendre.leggauss(nh)
        pts_hor = pts_hor * bounds
        pts_ver, w_ver = np.polynomial.legendre.leggauss(nv)
        pts_ver = pts_ver * bounds
        power_sq_pts = vector_1d_to_polar(angles, power_sq, pts_hor, pts_ver).T
        weights = power_sq_pts * np.outer(w_hor, w_ver)
        weights *= np.abs(np.cos(np.deg2rad(pts_ver)))
        weights *= 2 * bounds
        weights /= np.sum(weights.ravel())
        return Quadrature('grid', pts_hor, pts_ver, weights, _threshold(weights, config),
                          nh > 1 or nv > 1)
    raise NotImplementedError('integration scheme %r' % (scheme,))


def ml_edge_mask(mask_ml, n):
    """Per-gate factor of the low-weight sub-beams of scheme 'ml': deltas at the
    first / last melting-layer gate smoothed by scipy's gaussian_filter(sigma=2)
    (interpolation.py:423-436)."""
    from scipy.ndimage import gaussian_filter
    out = np.zeros(n)
    idx = np.where(mask_ml)[0] if mask_ml is not None else []
    if len(idx):
        out[idx[0]] = 1
        out[idx[-1]] = 1
        out = gaussian_filter(out, 2)
    return out
