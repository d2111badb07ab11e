"""Refraction scheme 2: ray paths from the Zeng & Blahak (2014) ODE.

Host-side, one solve per distinct (elevation + vertical quadrature node), exactly
as the reference does it -- scipy's LSODA over scipy's interp1d of the refractivity column,
on the dtypes the model fields come in (cosmo_pol/interpolation/atm_refraction.py:56-179):
the resulting float32 (s, h, e) tables are handed to the GPU through CPOL_GEOM_HOST_PATHS.  Quirks kept: the refractivity column is
read at [round(p0), round(p0)] (the row index is used for both axes, :113-117)
and the earth radius is evaluated with the radar LATITUDE in degrees fed to
cos/sin (:120, utilities.py:126-139).
"""
import os

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
    differences of the path, so the interpolant is the reference's own, not a re-derivation.
    Inside the table the value is interp1d's, evaluated without its per-call overhead (20 us of
    argument checking per scalar, two calls per right-hand side, ~1 500 right-hand sides per ray):
    the same statements on the same dtypes -- slope = (y_hi - y_lo) / (x_hi - x_lo) in the dtype of the
    column (float32 for model fields), then slope * (v - x_lo) + y_lo in float64, segment chosen by
    searchsorted(side='left') clipped to [1, n - 1] -- checked bit for bit against interp1d
    (tests/test_refraction_cpu.py)."""

    def __init__(self, x, y):
        from scipy.interpolate import interp1d
        self.f = interp1d(x, y)                 # (dtypes as given: a float32 column interpolates with float32 slopes)
        self.x, self.y = self.f.x, self.f.y
        x_, y_ = np.asarray(self.x), np.asarray(self.y)
        slope = (y_[1:] - y_[:-1]) / (x_[1:] - x_[:-1])          # dtype of the column, as interp1d forms it per call
        self._xs = [float(v) for v in x_]
        self._x_lo = self._xs[:-1]
        self._y_lo = [float(v) for v in y_[:-1]]
        self._slope = [float(v) for v in slope]
        self._n = len(self._xs)

    def __call__(self, v):
        x, y = self.x, self.y
        if v < x[0]:
            return y[0] + (v - x[0]) * (y[1] - y[0]) / (x[1] - x[0])
        if v > x[-1]:
            return y[-1] + (v - x[-1]) * (y[-1] - y[-2]) / (x[-1] - x[-2])
        import bisect
        i = min(max(bisect.bisect_left(self._xs, v), 1), self._n - 1) - 1
        return self._slope[i] * (v - self._x_lo[i]) + self._y_lo[i]


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


_SOLVED = {}            # (digest of column + range grid + site, elevation) -> [3, n_gates] float32; the last few scans
_SOLVED_MAX = 4096


def _digest(range_vec, coords_radar, h_col, n_col):
    import hashlib
    h = hashlib.blake2b(digest_size=16)
    for a in (range_vec, coords_radar, h_col, n_col):
        a = np.ascontiguousarray(a)
        h.update(str((a.dtype.str, a.shape)).encode())
        h.update(a.tobytes())
    return h.hexdigest()


def ode_paths(range_vec, elevations, pts_ver, coords_radar, h_col, n_col, workers=None):
    """float32 [n_rays, n_vnodes, 3, n_gates]; solves are shared between rays with equal elevation (a PPI
    needs n_vnodes solves) and kept across calls for the same refractivity column, range grid and site (a
    volume scanned again over the same model state costs nothing).  One solve takes 4 - 8 ms of host time
    (LSODA with a Python right-hand side, as in the reference), an RHI of 90 elevations with 3 vertical
    nodes 270 of them: from 16 missing solves on they are spread over `workers` helper processes
    (`python -m cosmo_pol_amd.refraction --worker`, started once and kept; default min(16, cores),
    CPOL_REFRACTION_WORKERS, 0 = solve here) -- the same function on the same inputs: the same bits."""
    el = np.asarray(elevations, dtype=np.float64)
    out = np.zeros((len(el), len(pts_ver), 3, len(range_vec)), dtype=np.float32)
    dg = _digest(range_vec, coords_radar, h_col, n_col)
    keys = sorted({float(pt + e) for e in el for pt in pts_ver})
    missing = [k for k in keys if (dg, k) not in _SOLVED]
    if workers is None:
        workers = int(os.environ.get('CPOL_REFRACTION_WORKERS', min(16, os.cpu_count() or 1)))
    solved = None
    if len(missing) >= 16 and workers > 1:
        solved = _pool_solve(range_vec, coords_radar, h_col, n_col, missing, workers)
    if solved is None:
        solved = [np.stack(ode_path(range_vec, k, coords_radar, h_col, n_col)) for k in missing]
    if len(_SOLVED) + len(missing) > _SOLVED_MAX:
        _SOLVED.clear()
    for k, v in zip(missing, solved):
        _SOLVED[(dg, k)] = v
    for r in range(len(el)):
        for j, pt in enumerate(pts_ver):
            out[r, j] = _SOLVED[(dg, float(pt + el[r]))]
    return out


# ---------------------------------------------------------------- helper processes
_POOL = []


def _pool_solve(range_vec, coords_radar, h_col, n_col, keys, workers):
    """Solves `keys` (elevations) in helper processes; None if they cannot be started (the caller solves here).
    Plain subprocesses speaking length-prefixed pickles over their pipes: no fork of a process that holds a GPU
    context, no re-import of the caller's __main__ (multiprocessing's spawn / forkserver would do that)."""
    import pickle
    import struct
    import subprocess
    import sys
    try:
        while len(_POOL) < workers:
            root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
            env = dict(os.environ, PYTHONPATH=root + os.pathsep + os.environ.get('PYTHONPATH', ''),
                       OMP_NUM_THREADS='1', OPENBLAS_NUM_THREADS='1', MKL_NUM_THREADS='1')
            _POOL.append(subprocess.Popen([sys.executable, '-m', 'cosmo_pol_amd.refraction', '--worker'],
                                          stdin=subprocess.PIPE, stdout=subprocess.PIPE, env=env))
        if not _POOL_ATEXIT:
            import atexit
            atexit.register(_pool_close)
            _POOL_ATEXIT.append(1)
        parts = [keys[i::workers] for i in range(workers)]
        used = []
        for proc, part in zip(_POOL, parts):
            if not part:
                continue
            msg = pickle.dumps((np.asarray(range_vec), list(coords_radar), np.asarray(h_col), np.asarray(n_col), part), protocol=4)
            proc.stdin.write(struct.pack('<Q', len(msg)) + msg)
            proc.stdin.flush()
            used.append((proc, part))
        got = {}
        for proc, part in used:
            head = proc.stdout.read(8)
            if len(head) != 8:
                raise IOError('refraction helper process ended')
            n = struct.unpack('<Q', head)[0]
            for k, v in zip(part, pickle.loads(proc.stdout.read(n))):
                got[k] = v
        return [got[k] for k in keys]
    except (OSError, IOError, ValueError, pickle.PickleError):
        _pool_close()
        return None


_POOL_ATEXIT = []


def _pool_close():
    while _POOL:
        p = _POOL.pop()
        try:
            p.stdin.close()
            p.wait(timeout=2)
        except Exception:
            p.kill()


def _worker_main():
    import pickle
    import struct
    import sys
    inp, out = sys.stdin.buffer, sys.stdout.buffer
    while True:
        head = inp.read(8)
        if len(head) != 8:
            return
        rv, coords, h_col, n_col, keys = pickle.loads(inp.read(struct.unpack('<Q', head)[0]))
        res = pickle.dumps([np.stack(ode_path(rv, k, coords, h_col, n_col)) for k in keys], protocol=4)
        out.write(struct.pack('<Q', len(res)) + res)
        out.flush()


if __name__ == '__main__':
    import sys
    if '--worker' in sys.argv:
        _worker_main()
