"""Antenna-pattern quadratures: the (azimuth offset, elevation offset, weight)
sub-beams of a radial, for every `integration/scheme` the reference's own
configuration check accepts except scheme 4 (cfg.py:104):

  1     Gauss-Hermite on a single Gaussian beam   interpolation/interpolation.py:148-166
  'ml'  Gauss-Hermite with 10x more vertical nodes, sorted by weight; the nodes
        beyond the first nv_GH + 1 only contribute near the two edges of the
        melting layer (per-gate weights, built on the GPU)        :168-193, 423-436
  2     sum of Gaussians fitted to the antenna diagram, Gauss-Hermite (radial) x
        Gauss-Legendre (angular) in polar coordinates -> point list  :195-229
  3     Gauss-Legendre on +-max(angle) weighted by the measured two-way antenna
        diagram                                                     :231-256
  kept sub-beams: threshold on the cumulated sorted weights         :342-354, 383

Scheme 4 reads a pickle from a hard-coded path of the authors' disk (:258-262), schemes
5 and 6 are rejected by the reference's own VALID_VALUES; they raise here.

This is cold-path host code (once per configuration): it produces the small
tables (`SubBeams`) the sweep kernels consume.
"""
import numpy as np

from .geometry import SubBeams

FWHM_TO_SIGMA = 1.0 / (2 * np.sqrt(2 * np.log(2)))
ML_FILTER_SIGMA = 2.0          # gaussian_filter(smoothing_mask, 2), interpolation.py:432
ML_FILTER_TRUNCATE = 4.0       # scipy's default


def kept_mask(weights, weight_threshold, always=False):
    """weights >= the weight at which the cumulated, descending-sorted, normalised
    weights reach `weight_threshold`; the kept weights are NOT renormalised."""
    flat = np.sort(np.array(weights).ravel())[::-1]
    cum = np.cumsum(flat / np.sum(flat))
    cum[-1] = 1.
    thr = flat[np.where(cum >= weight_threshold)[0][0]]
    keep = np.asarray(weights) >= thr
    if always:
        keep[...] = True
    return keep


def _hermite_grid(config, n_ver, sort_vertical):
    bw = config['radar']['3dB_beamwidth']
    nh = int(config['integration']['nh_GH'])
    sigma = bw * FWHM_TO_SIGMA
    x_h, w_h = np.polynomial.hermite.hermgauss(nh)
    x_v, w_v = np.polynomial.hermite.hermgauss(n_ver)
    x_h, x_v = x_h * sigma, x_v * sigma
    if sort_vertical:
        order = np.argsort(w_v)[::-1]
        w_v, x_v = w_v[order], x_v[order]
    W = np.outer(w_h * sigma, w_v * sigma)
    W *= np.abs(np.cos(np.deg2rad(x_v)))
    W /= np.sum(W.ravel())
    return x_h, x_v, W


def antenna_power_sq(path):
    """(angles, two-way power) of a comma-separated antenna diagram file
    (angle in degrees, one-way power in dB)."""
    a = np.genfromtxt(path, delimiter=',')
    return a[:, 0], (10 ** (0.1 * a[:, 1])) ** 2


def diagram_on_grid(angles, values, x, y):
    """Rotates the 1-D cut `values(angles)` around the beam axis and samples it at the
    Cartesian offsets x (horizontal) x y (vertical): utilities.py:285-341
    (vector_1d_to_polar / polar2cartesian, cubic map_coordinates)."""
    from scipy.interpolate import interp1d
    from scipy.ndimage import map_coordinates
    mid = int(np.floor(len(angles) / 2.))
    r = angles[mid:]
    t = [0, np.pi, 2 * np.pi]
    pol = np.zeros((3, len(r)))
    pol[0, :] = values[mid:]
    pol[1, :] = values[0:mid + 1][::-1]
    pol[2, :] = pol[0, :]
    X, Y = np.meshgrid(x, y)
    rr = np.sqrt(X * X + Y * Y)
    tt = np.arctan2(X, Y) + np.pi
    i_r = interp1d(r, np.arange(len(r)), bounds_error=False)(rr.ravel())
    i_t = interp1d(t, np.arange(len(t)))(tt.ravel())
    i_r[rr.ravel() > r.max()] = len(r) - 1
    i_r[rr.ravel() < r.min()] = 0
    return map_coordinates(pol, np.array([i_t, i_r]), order=3).reshape(rr.shape)


def fit_gaussians(angles, power_db, n_gaussians):
    """Sum-of-Gaussians fit of an antenna diagram (antenna_fit.py:74-112): start from
    the strongest lobes, SLSQP on the dB residual, main lobe pinned at 0 dB / 0 deg.
    Rows of the result: (amplitude dB, offset deg, sigma deg)."""
    from scipy.optimize import minimize
    from scipy.signal import argrelextrema
    peaks = argrelextrema(power_db, np.greater)
    amp, mu = power_db[peaks], angles[peaks]
    if 0 not in mu:
        mu, amp = np.append(mu, 0), np.append(amp, 0)
    lobes = np.column_stack((amp, mu))
    lobes = np.flipud(lobes[lobes[:, 0].argsort()])[0:n_gaussians, :]
    p0 = np.column_stack((lobes[:, 0], lobes[:, 1], np.full(n_gaussians, 0.5)))
    bounds = []
    for _ in range(n_gaussians):
        bounds += [[None, None], [None, None], [0.1, 2]]
    bounds[0] = [0, 0]
    bounds[1] = [0, 0]

    def cost(p):
        p = np.reshape(p, (len(p) // 3, 3))
        est = 10 * np.log10(np.sum([10 ** (0.1 * a) * np.exp(-(angles - m) ** 2 / (2 * s ** 2))
                                    for a, m, s in p], axis=0))
        return np.sqrt(np.sum((est - power_db) ** 2))

    res = minimize(cost, p0.ravel(), bounds=bounds, method='SLSQP')
    return np.reshape(res['x'], (n_gaussians, 3))


def ml_filter_taps():
    """Taps of scipy.ndimage.gaussian_filter(., 2) (radius 8), centre first."""
    radius = int(ML_FILTER_TRUNCATE * ML_FILTER_SIGMA + 0.5)
    x = np.arange(-radius, radius + 1)
    phi = np.exp(-0.5 / (ML_FILTER_SIGMA * ML_FILTER_SIGMA) * x ** 2)
    phi = phi / phi.sum()
    return radius, np.ascontiguousarray(phi, dtype=np.float64)


def subbeams(config):
    integ = config['integration']
    scheme = integ['scheme']
    thr = integ['weight_threshold']
    if scheme == 1 or scheme == 'ml':
        nh, nv = int(integ['nh_GH']), int(integ['nv_GH'])
        n_ver = nv
        if scheme == 'ml':
            n_ver = 10 * nv
            if n_ver % 2 == 0:
                n_ver += 1
        x_h, x_v, W = _hermite_grid(config, n_ver, scheme == 'ml')
        sb = SubBeams(x_h, x_v, W, kept_mask(W, thr, always=not (nh > 1 or nv > 1)))
        if scheme == 'ml':
            sb.sub_smooth = np.ascontiguousarray(sb.sub_v > nv, dtype=np.int32)
            sb.ml_radius, sb.ml_filter = ml_filter_taps()
        return sb
    if scheme == 3:
        if not integ.get('antenna_diagram'):
            raise ValueError('integration/scheme 3 needs integration/antenna_diagram')
        nh, nv = int(integ['nh_GH']), int(integ['nv_GH'])
        angles, p2 = antenna_power_sq(integ['antenna_diagram'])
        half = np.max(angles)
        x_h, w_h = np.polynomial.legendre.leggauss(nh)
        x_v, w_v = np.polynomial.legendre.leggauss(nv)
        x_h, x_v = x_h * half, x_v * half
        W = diagram_on_grid(angles, p2, x_h, x_v).T * np.outer(w_h, w_v)
        W *= np.abs(np.cos(np.deg2rad(x_v)))
        W *= 2 * half
        W /= np.sum(W.ravel())
        return SubBeams(x_h, x_v, W, kept_mask(W, thr, always=not (nh > 1 or nv > 1)))
    if scheme == 2:
        ap = integ.get('antenna_params')
        if ap is None:
            raise ValueError('integration/scheme 2 needs integration/antenna_diagram (fitted at '
                             'configuration time) or integration/antenna_params')
        ap = np.asarray(ap, dtype=np.float64)
        x_a, w_a = np.polynomial.legendre.leggauss(int(integ['na_GL']))
        x_r, w_r = np.polynomial.hermite.hermgauss(int(integ['nr_GH']))
        off_h, off_v, w = [], [], []
        total = 0
        for i in range(len(x_r)):
            for a_db, mu, sig in ap:
                for k in range(len(x_a)):
                    r = mu + np.sqrt(2) * sig * x_r[i]
                    theta = np.pi * x_a[k] + np.pi
                    wk = (np.pi * w_a[k] * w_r[i] * 10 ** (0.1 * a_db) * np.sqrt(2) * sig * abs(r))
                    wk *= np.cos(r * np.sin(theta))
                    w.append(wk)
                    total += wk
                    off_h.append(r * np.cos(theta))
                    off_v.append(r * np.sin(theta))
        w = np.array(w)
        w /= total
        keep = kept_mask(w, thr)
        return SubBeams.from_points(np.array(off_h)[keep], np.array(off_v)[keep], w[keep])
    if scheme == 4:
        raise NotImplementedError('integration/scheme 4 reads the authors\' own antenna pickle '
                                  '(interpolation.py:258-262) and cannot be reproduced')
    raise NotImplementedError('integration/scheme %r is not accepted by the reference '
                              'configuration either (cfg.py:104)' % (scheme,))
