"""Oracle GPM-DPR swath geometry (TEST INFRASTRUCTURE ONLY).

The reference's GPM branch cannot run as shipped, so there is
no reference output to pin this module: it restates the INTENDED behaviour of
  get_GPM_angles          cosmo_pol/radar/gpm_wrapper.py:259-322
  compute_trajectory_GPM  cosmo_pol/interpolation/atm_refraction.py:222-272
      (h < MAX_MODEL_HEIGHT filter with a proper boolean mask, the elevation kept
       in degrees -- the second rad2deg at :270 is a bug --, earth radius from
       the satellite latitude through get_earth_radius, quirk Q1)
  GPM worker              cosmo_pol/radar_operator.py:603-637 (per-ray radar site)
PARITY UNPINNED against the reference for config 5; pyproj's Geod.inv /
geocentric transform are replaced by Vincenty's inverse formula and the
closed-form WGS84 geodetic -> ECEF conversion.
"""
import numpy as np

from . import beam
from . import constants as K
from .geodesy import DEG, WGS84_A, WGS84_F


def geodetic_to_ecef(lat_deg, lon_deg, h):
    lat, lon = np.deg2rad(lat_deg), np.deg2rad(lon_deg)
    e2 = 2 * WGS84_F - WGS84_F ** 2
    N = WGS84_A / np.sqrt(1 - e2 * np.sin(lat) ** 2)
    return ((N + h) * np.cos(lat) * np.cos(lon), (N + h) * np.cos(lat) * np.sin(lon),
            (N * (1 - e2) + h) * np.sin(lat))


def inverse_azimuth(lat1, lon1, lat2, lon2):
    """Vincenty inverse, forward azimuth in degrees (scalar inputs)."""
    f = WGS84_F
    U1 = np.arctan((1 - f) * np.tan(lat1 * DEG))
    U2 = np.arctan((1 - f) * np.tan(lat2 * DEG))
    L = (lon2 - lon1) * DEG
    lam = L
    for _ in range(200):
        ss = np.hypot(np.cos(U2) * np.sin(lam),
                      np.cos(U1) * np.sin(U2) - np.sin(U1) * np.cos(U2) * np.cos(lam))
        if ss == 0:
            return 0.0
        cs = np.sin(U1) * np.sin(U2) + np.cos(U1) * np.cos(U2) * np.cos(lam)
        sig = np.arctan2(ss, cs)
        sa = np.cos(U1) * np.cos(U2) * np.sin(lam) / ss
        c2a = 1 - sa * sa
        c2sm = cs - 2 * np.sin(U1) * np.sin(U2) / c2a if c2a != 0 else 0.0
        C = f / 16 * c2a * (4 + f * (4 - 3 * c2a))
        new = L + (1 - C) * f * sa * (sig + C * ss * (c2sm + C * cs * (-1 + 2 * c2sm ** 2)))
        if abs(new - lam) < 1e-15:
            lam = new
            break
        lam = new
    return np.rad2deg(np.arctan2(np.cos(U2) * np.sin(lam),
                                 np.cos(U1) * np.sin(U2) - np.sin(U1) * np.cos(U2) * np.cos(lam)))


def swath_angles(swath):
    lat2, lon2 = swath['Latitude'], swath['Longitude']
    N, M = lat2.shape
    az = np.zeros((N, M))
    el = np.zeros((N, M))
    rng = np.zeros((N, M))
    for i in range(N):
        pos = swath['scPos'][i]
        H = np.sqrt(np.sum(pos ** 2))
        RE = H - swath['dprAlt'][i]
        for j in range(M):
            az[i, j] = inverse_azimuth(swath['scLat'][i], swath['scLon'][i], lat2[i, j], lon2[i, j])
            x, y, z = geodetic_to_ecef(lat2[i, j], lon2[i, j], 0.0)
            r = np.sqrt((x - pos[0]) ** 2 + (y - pos[1]) ** 2 + (z - pos[2]) ** 2)
            rng[i, j] = r
            with np.errstate(invalid='ignore'):
                theta = -np.arcsin((H ** 2 + r ** 2 - RE ** 2) / (2 * H * r)) / np.pi * 180.
            if np.isnan(theta):
                theta = -90
            el[i, j] = -theta
    coords = np.vstack((swath['scLat'], swath['scLon'], swath['dprAlt'])).T
    return az, el, rng, coords


def spaceborne_heights(range_vec, elevation_deg, coords):
    el = np.deg2rad(elevation_deg)
    KE = 1
    RE = beam.earth_radius(coords[0])
    return -(np.sqrt(range_vec ** 2 + (KE * RE) ** 2 + 2 * range_vec * KE * RE * np.sin(el))
             - KE * RE) + coords[2]


def spaceborne_window(elevation_deg, coords, max_range, res):
    """(first kept candidate gate, number of kept gates) of the central ray."""
    rv = np.arange(res / 2., max_range, res)
    h = spaceborne_heights(rv, elevation_deg, coords)
    low = np.where(h < K.MAX_MODEL_HEIGHT)[0]
    k0 = int(low[0]) if len(low) else len(rv)
    return k0, len(rv) - k0


def trajectory_spaceborne(elevation_deg, coords, res, k0, n):
    rv = res / 2. + (k0 + np.arange(n)) * float(res)
    el = np.deg2rad(elevation_deg)
    KE = 1
    RE = beam.earth_radius(coords[0])
    alt = coords[2]
    h = spaceborne_heights(rv, elevation_deg, coords)
    s = KE * RE * np.arcsin((rv * np.cos(el)) / (KE * RE + h))
    e = elevation_deg - np.rad2deg(np.arctan(rv * np.cos(el) / (rv * np.sin(el) + KE * RE + alt)))
    return s.astype('float32'), h.astype('float32'), e.astype('float32')


def interpolate_swath_ray(cube, config, azimuth, elevation, slant_range, coords):
    """Sub-beams of one swath ray; all vertical nodes share the central node's
    gate window (aligned ranges)."""
    res = config['radar']['radial_resolution']
    pts_hor, pts_ver, weights, keep = beam.gauss_hermite_subbeams(config)
    jc = int(len(pts_ver) / 2)
    k0, n = spaceborne_window(pts_ver[jc] + elevation, coords, slant_range, res)
    trajs = [trajectory_spaceborne(pt + elevation, coords, res, k0, n) for pt in pts_ver]
    return beam.interpolate_radial(cube, config, azimuth, elevation, trajs=trajs,
                                   coords_radar=coords), k0, n
