"""GPM-DPR swath geometry and output container (BASELINE config 5).

The reference's GPM branch is dead as shipped (`update_config`
does not exist, `get_GPM_angles` reads an undefined name, `compute_trajectory_GPM`
reads a non-existent constant and applies rad2deg twice).  This module restates
the INTENDED behaviour:

  get_GPM_angles          cosmo_pol/radar/gpm_wrapper.py:259-322  -- for every swath
                          pixel: azimuth of the WGS84 geodesic satellite -> pixel,
                          slant range to the pixel at h = 0 (ECEF), elevation from
                          the law of cosines in the triangle earth centre /
                          satellite / pixel (positive number, ray points down)
  compute_trajectory_GPM  cosmo_pol/interpolation/atm_refraction.py:222-272 -- runs
                          on the GPU (csrc/cpol_interp.inl, CPOL_GEOM_SPACEBORNE)
  band constants          cosmo_pol/constants/global_constants.py:152-159,
                          cosmo_pol/radar_operator.py:577-588
  SimulatedGPM            cosmo_pol/radar/gpm_wrapper.py:47-141

pyproj (Geod.inv, geocentric projection) and h5py are not installable here: the
inverse geodesic is Vincenty's inverse formula and the ECEF conversion the
closed-form WGS84 expression ("parity unpinned", like the direct problem).  No
reference output can pin config 5; parity is defined against the oracle's own
restatement (tests/test_gpu_gpm.py).
"""
import numpy as np

from . import constants as K
from .geometry import DEG, WGS84_A, WGS84_B, WGS84_F


def lla_to_ecef(lat_deg, lon_deg, h):
    lat = np.asarray(lat_deg, dtype=np.float64) * DEG
    lon = np.asarray(lon_deg, dtype=np.float64) * DEG
    e2 = WGS84_F * (2.0 - WGS84_F)
    n = WGS84_A / np.sqrt(1.0 - e2 * np.sin(lat) ** 2)
    x = (n + h) * np.cos(lat) * np.cos(lon)
    y = (n + h) * np.cos(lat) * np.sin(lon)
    z = (n * (1.0 - e2) + h) * np.sin(lat)
    return x, y, z


def wgs84_inverse_azimuth(lat1_deg, lon1_deg, lat2_deg, lon2_deg, iters=30):
    """Forward azimuth [deg] of the WGS84 geodesic 1 -> 2 (Vincenty inverse)."""
    f = WGS84_F
    phi1 = np.asarray(lat1_deg, dtype=np.float64) * DEG
    phi2 = np.asarray(lat2_deg, dtype=np.float64) * DEG
    L = (np.asarray(lon2_deg, dtype=np.float64) - np.asarray(lon1_deg, dtype=np.float64)) * DEG
    U1 = np.arctan((1 - f) * np.tan(phi1))
    U2 = np.arctan((1 - f) * np.tan(phi2))
    sU1, cU1, sU2, cU2 = np.sin(U1), np.cos(U1), np.sin(U2), np.cos(U2)
    lam = L + 0.0 * U2
    for _ in range(iters):
        sl, cl = np.sin(lam), np.cos(lam)
        sin_sigma = np.sqrt((cU2 * sl) ** 2 + (cU1 * sU2 - sU1 * cU2 * cl) ** 2)
        cos_sigma = sU1 * sU2 + cU1 * cU2 * cl
        sigma = np.arctan2(sin_sigma, cos_sigma)
        with np.errstate(invalid='ignore', divide='ignore'):
            sin_alpha = np.where(sin_sigma == 0, 0.0, cU1 * cU2 * sl / sin_sigma)
            cos2_alpha = 1 - sin_alpha ** 2
            cos2sm = np.where(cos2_alpha == 0, 0.0, cos_sigma - 2 * sU1 * sU2 / cos2_alpha)
        C = f / 16 * cos2_alpha * (4 + f * (4 - 3 * cos2_alpha))
        lam = L + (1 - C) * f * sin_alpha * (sigma + C * sin_sigma * (
            cos2sm + C * cos_sigma * (-1 + 2 * cos2sm ** 2)))
    sl, cl = np.sin(lam), np.cos(lam)
    az = np.arctan2(cU2 * sl, cU1 * sU2 - sU1 * cU2 * cl) / DEG
    return az


def band_settings(band):
    """(frequency GHz, radial resolution m) (radar_operator.py:577-588)."""
    if band in ('Ku', 'Ku_matched'):
        freq = K.GPM_KU_FREQUENCY
    elif band in ('Ka', 'Ka_matched'):
        freq = K.GPM_KA_FREQUENCY
    else:
        raise ValueError("band must be 'Ku', 'Ka', 'Ku_matched' or 'Ka_matched'")
    res = K.GPM_RADIAL_RES_KA if band == 'Ka' else K.GPM_RADIAL_RES_KU
    return freq, res


def hdf5_group(band):
    return {'Ku': 'NS', 'Ka': 'HS', 'Ku_matched': 'MS', 'Ka_matched': 'MS'}[band]


def read_swath(GPM_file, band):
    """Swath description: dict with Latitude, Longitude [N, M], scLat, scLon,
    dprAlt [N], scPos [N, 3] (ECEF, m).  `GPM_file` is such a dict or the path of
    a GPM-DPR HDF5 file (needs h5py)."""
    if isinstance(GPM_file, dict):
        return {k: np.asarray(GPM_file[k], dtype=np.float64)
                for k in ('Latitude', 'Longitude', 'scLat', 'scLon', 'dprAlt', 'scPos')}
    try:
        import h5py
    except ImportError:
        raise ImportError('reading GPM-DPR HDF5 files needs h5py, which is not available here; '
                          'pass a dict with Latitude, Longitude, scLat, scLon, dprAlt, scPos')
    with h5py.File(GPM_file, 'r') as f:
        g = f[hdf5_group(band)]
        nav = g['navigation']
        return dict(Latitude=g['Latitude'][:], Longitude=g['Longitude'][:], scLat=nav['scLat'][:],
                    scLon=nav['scLon'][:], dprAlt=nav['dprAlt'][:], scPos=nav['scPos'][:])


def swath_angles(swath):
    """azimuths, elevations, slant ranges [N, M] and satellite coordinates [N, 3]
    (lat, lon, altitude) (gpm_wrapper.py:272-322)."""
    lat2, lon2 = swath['Latitude'], swath['Longitude']
    N, M = lat2.shape
    sc_lat = np.broadcast_to(swath['scLat'][:, None], (N, M))
    sc_lon = np.broadcast_to(swath['scLon'][:, None], (N, M))
    az = wgs84_inverse_azimuth(sc_lat, sc_lon, lat2, lon2)
    sx, sy, sz = lla_to_ecef(lat2, lon2, 0.0)
    pos = swath['scPos']
    rng = np.sqrt((sx - pos[:, None, 0]) ** 2 + (sy - pos[:, None, 1]) ** 2
                  + (sz - pos[:, None, 2]) ** 2)
    H = np.sqrt(pos[:, 0] ** 2 + pos[:, 1] ** 2 + pos[:, 2] ** 2)[:, None]
    RE = H - swath['dprAlt'][:, None]
    with np.errstate(invalid='ignore'):
        theta = -np.arcsin((H ** 2 + rng ** 2 - RE ** 2) / (2 * H * rng)) / np.pi * 180.
    theta = np.where(np.isnan(theta), -90.0, theta)
    coords = np.stack([swath['scLat'], swath['scLon'], swath['dprAlt']], axis=1)
    return az, -theta, rng, coords


def synthetic_swath(n_scans=20, n_rays=49, centre=(46.5, 7.5), heading_deg=20.0, altitude=407000.0,
                    cross_track_deg=17.0, scan_spacing_m=5000.0, seed=0):
    """A seeded synthetic DPR-like swath over the given point (SURVEY 8(d):
    407 km orbit, +-17 deg cross-track) in the dict format of read_swath."""
    lat0, lon0 = centre
    # ground track: geodesic through the centre with the given heading (spherical step)
    d = (np.arange(n_scans) - (n_scans - 1) / 2.0) * scan_spacing_m
    R = 6371000.0
    sc_lat = lat0 + (d * np.cos(heading_deg * DEG) / R) / DEG
    sc_lon = lon0 + (d * np.sin(heading_deg * DEG) / (R * np.cos(lat0 * DEG))) / DEG
    alt = np.full(n_scans, float(altitude))
    px, py, pz = lla_to_ecef(sc_lat, sc_lon, alt)
    sc_pos = np.stack([px, py, pz], axis=1)
    # cross-track pixels: ground offset of a beam tilted by the scan angle
    ang = np.linspace(-cross_track_deg, cross_track_deg, n_rays) * DEG
    ground = altitude * np.tan(ang)                       # flat-earth offset is enough here
    perp = (heading_deg + 90.0) * DEG
    lat = sc_lat[:, None] + (ground[None, :] * np.cos(perp) / R) / DEG
    lon = sc_lon[:, None] + (ground[None, :] * np.sin(perp) / (R * np.cos(lat0 * DEG))) / DEG
    return dict(Latitude=lat, Longitude=lon, scLat=sc_lat, scLon=sc_lon, dprAlt=alt, scPos=sc_pos)


class SimulatedGPM(object):
    """Output of get_GPM_swath (gpm_wrapper.py:47-141): gates below the model
    topography are dropped, every beam is flipped to start at the ground and
    packed into [N, M, max_len] arrays."""

    def __init__(self, fields, mask, lats, lons, n_kept, dim, band):
        """`data` is a LazyDict and `lats` / `lons` / `bin_surface` are built on first access: every packed array
        is a float64 [N, M, max_len] scatter of ~3 M values (8 ms each on the host), eleven of them per swath;
        the boolean passes over the gates behind `bin_surface` alone were a third of a Ku swath call (3.3 of 10 ms)."""
        from .radar_operator import LazyDict
        N, M = dim
        n_rays, n_gates = mask.shape
        self.band = band
        n_kept = np.asarray(n_kept).reshape(-1)
        idx = {}

        def gates_of_ray():
            if 'inside' not in idx:
                idx['inside'] = np.arange(n_gates)[None, :] < n_kept[:, None]
            return idx['inside']

        def bin_surface():
            above = gates_of_ray() & (mask >= 1)
            first_above = np.where(above.any(axis=1), above.argmax(axis=1), -1)
            return np.where(first_above >= 0, n_kept - first_above, 0).astype(float).reshape(N, M)

        def indices():
            if 'src' not in idx:
                # kept gates (not below the topography), flipped so that index 0 is the lowest one
                keep = gates_of_ray() & (mask > -1)
                n = keep.sum(axis=1)
                src_idx = np.flatnonzero(keep)                           # flat source index of every kept gate
                dest = (n[:, None] - np.cumsum(keep, axis=1, dtype=np.int32)).reshape(-1)[src_idx]
                idx['src'] = src_idx
                idx['dst'] = (src_idx // n_gates) * n_gates + dest       # flat destination (beam flipped)
            return idx['src'], idx['dst']

        def pack(src, fill):
            src_idx, dst_idx = indices()
            out = np.full(n_rays * n_gates, fill, dtype=np.float64)
            out[dst_idx] = np.asarray(src).reshape(-1)[src_idx]
            return out.reshape(N, M, n_gates)
        self._coords = LazyDict()
        self._coords.add('bin_surface', bin_surface)
        self._coords.add('lats', lambda: pack(lats, np.nan))
        self._coords.add('lons', lambda: pack(lons, np.nan))
        self.data = LazyDict()
        for k in fields:
            self.data.add(k, (lambda kk: (lambda: pack(fields[kk], 0.0)))(k))

    @property
    def bin_surface(self):
        return self._coords['bin_surface']

    @property
    def lats(self):
        return self._coords['lats']

    @property
    def lons(self):
        return self._coords['lons']
