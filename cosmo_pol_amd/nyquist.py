"""Per-ray Nyquist velocity from a scan-strategy table (RVEL aliasing).

Restates cosmo_pol/config/nyquist_fit.py:14-30: a comma-separated file with the
columns elevation, azimuth, nyquist; the value of the row with the closest
elevation and, among those, the closest azimuth is used.  (The reference's
azimuth selection indexes a pandas Series with azimuth *values* and only works
for its shipped all-zero-azimuth file; the intended nearest-azimuth rule is
implemented here.)  The folding itself (cosmo_pol/utilities/utilities.py:142-156)
runs on the GPU (csrc/cpol_final.inl).
"""
import numpy as np


class NyquistTable(object):
    def __init__(self, filename):
        data = np.genfromtxt(filename, delimiter=',', names=True)
        self.elevation = np.atleast_1d(data['elevation']).astype(np.float64)
        self.azimuth = np.atleast_1d(data['azimuth']).astype(np.float64)
        self.nyquist = np.atleast_1d(data['nyquist']).astype(np.float64)
        self.filename = filename

    def __call__(self, elevation, azimuth):
        el = np.atleast_1d(np.asarray(elevation, dtype=np.float64))
        az = np.atleast_1d(np.asarray(azimuth, dtype=np.float64))
        out = np.empty(len(el))
        for i in range(len(el)):
            closest_el = self.elevation[np.argmin(np.abs(self.elevation - el[i]))]
            idx = np.where(self.elevation == closest_el)[0]
            j = idx[np.argmin(np.abs(self.azimuth[idx] - az[i]))]
            out[i] = self.nyquist[j]
        return out
