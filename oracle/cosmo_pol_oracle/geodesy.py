"""Oracle geodesy (TEST INFRASTRUCTURE ONLY; see oracle/README.md).

Stands in for the two third-party calls on the reference's hot path whose
source is NOT under /root/reference:

  * pyproj.Geod(ellps='WGS84').fwd(lon, lat, az, dist)
        reference call site: cosmo_pol/interpolation/interpolation.py:526,534
        (pyproj unpinned in setup.py:50) -> restated as Vincenty's (1975)
        direct formula on the WGS84 ellipsoid with a FIXED number of
        iterations (so that the HIP kernel can mirror the exact operation
        sequence; differences with Karney's algorithm used by PROJ are
        < 0.1 mm, i.e. < 1e-9 deg).
  * pycosmo.WGS_to_COSMO((lats, lons), [SP_lat, SP_lon])
        reference call sites: interpolation.py:566-568, atm_refraction.py:101
        (pycosmo unpinned, not in install_requires) -> restated as the standard
        rotated-pole rotation; output [n,2] float32, col 0 = rotated latitude,
        col 1 = rotated longitude (contract inferred from interpolation.py:
        572-575 and interpolation_c.c:43-44).

PARITY UNPINNED against the real pyproj / pycosmo (neither installed nor
vendored); pinned instead by tests/test_geodesy_kat_cpu.py against an independent
numerical integration of the geodesic ODE and a literature known answer.
"""
import numpy as np

WGS84_A = 6378137.0
WGS84_F = 1.0 / 298.257223563
WGS84_B = (1.0 - WGS84_F) * WGS84_A
VINCENTY_ITERS = 5
DEG = np.pi / 180.0


def direct_ray_constants(lat1_deg, lon1_deg, az_deg):
    """Per-ray constants of Vincenty's direct problem (host side of the
    product mirrors this function; the per-gate part is the kernel)."""
    f = WGS84_F
    a = WGS84_A
    b = WGS84_B
    phi1 = np.float64(lat1_deg) * DEG
    alpha1 = np.asarray(az_deg, dtype=np.float64) * DEG
    sin_a1 = np.sin(alpha1)
    cos_a1 = np.cos(alpha1)
    tan_u1 = (1.0 - f) * np.tan(phi1)
    cos_u1 = 1.0 / np.sqrt(1.0 + tan_u1 * tan_u1)
    sin_u1 = tan_u1 * cos_u1
    sigma1 = np.arctan2(tan_u1, cos_a1)
    sin_alpha = cos_u1 * sin_a1
    cos2_alpha = 1.0 - sin_alpha * sin_alpha
    u2 = cos2_alpha * (a * a - b * b) / (b * b)
    A = 1.0 + u2 / 16384.0 * (4096.0 + u2 * (-768.0 + u2 * (320.0 - 175.0 * u2)))
    B = u2 / 1024.0 * (256.0 + u2 * (-128.0 + u2 * (74.0 - 47.0 * u2)))
    C = f / 16.0 * cos2_alpha * (4.0 + f * (4.0 - 3.0 * cos2_alpha))
    return dict(sin_a1=sin_a1, cos_a1=cos_a1, sin_u1=sin_u1, cos_u1=cos_u1,
                sigma1=sigma1, sin_alpha=sin_alpha, cos2_alpha=cos2_alpha,
                bA=b * A, B=B, C=C)


def direct_gate(k, lon1_deg, s):
    """Per-gate part of Vincenty direct; `k` from direct_ray_constants,
    `s` ground distances [m] (float64). Returns (lat_deg, lon_deg)."""
    f = WGS84_F
    s = np.asarray(s, dtype=np.float64)
    sigma0 = s / k["bA"]
    sigma = sigma0
    B = k["B"]
    for _ in range(VINCENTY_ITERS):
        two_sm = 2.0 * k["sigma1"] + sigma
        cos2sm = np.cos(two_sm)
        sin_s = np.sin(sigma)
        cos_s = np.cos(sigma)
        dsig = B * sin_s * (cos2sm + B / 4.0 * (cos_s * (-1.0 + 2.0 * cos2sm * cos2sm)
                            - B / 6.0 * cos2sm * (-3.0 + 4.0 * sin_s * sin_s)
                            * (-3.0 + 4.0 * cos2sm * cos2sm)))
        sigma = sigma0 + dsig
    two_sm = 2.0 * k["sigma1"] + sigma
    cos2sm = np.cos(two_sm)
    sin_s = np.sin(sigma)
    cos_s = np.cos(sigma)
    tmp = k["sin_u1"] * sin_s - k["cos_u1"] * cos_s * k["cos_a1"]
    lat2 = np.arctan2(k["sin_u1"] * cos_s + k["cos_u1"] * sin_s * k["cos_a1"],
                      (1.0 - f) * np.sqrt(k["sin_alpha"] * k["sin_alpha"] + tmp * tmp))
    lam = np.arctan2(sin_s * k["sin_a1"],
                     k["cos_u1"] * cos_s - k["sin_u1"] * sin_s * k["cos_a1"])
    C = k["C"]
    L = lam - (1.0 - C) * f * k["sin_alpha"] * (
        sigma + C * sin_s * (cos2sm + C * cos_s * (-1.0 + 2.0 * cos2sm * cos2sm)))
    lon2 = np.float64(lon1_deg) + L / DEG
    return lat2 / DEG, lon2


def wgs84_direct(lat1_deg, lon1_deg, az_deg, s):
    """WGS84 direct geodesic: start (lat1, lon1) [deg], azimuth [deg], ground
    distances s [m] -> (lat2, lon2) [deg]."""
    k = direct_ray_constants(lat1_deg, lon1_deg, az_deg)
    return direct_gate(k, lon1_deg, s)


def rotation_constants(sp_lat_deg, sp_lon_deg):
    theta = (90.0 + np.float64(sp_lat_deg)) * DEG
    phi = np.float64(sp_lon_deg) * DEG
    return dict(ct=np.cos(theta), st=np.sin(theta), cp=np.cos(phi), sp=np.sin(phi))


def wgs_to_rotated(lats_deg, lons_deg, sp_lat_deg, sp_lon_deg):
    """Geographic -> rotated-pole coordinates given the rotated SOUTH pole.
    Returns float32 [n, 2]: col 0 = rotated lat, col 1 = rotated lon."""
    r = rotation_constants(sp_lat_deg, sp_lon_deg)
    lat = np.asarray(lats_deg, dtype=np.float64) * DEG
    lon = np.asarray(lons_deg, dtype=np.float64) * DEG
    cl = np.cos(lat)
    x = np.cos(lon) * cl
    y = np.sin(lon) * cl
    z = np.sin(lat)
    x_new = r["ct"] * r["cp"] * x + r["ct"] * r["sp"] * y + r["st"] * z
    y_new = -r["sp"] * x + r["cp"] * y
    z_new = -r["st"] * r["cp"] * x - r["st"] * r["sp"] * y + r["ct"] * z
    rlon = np.arctan2(y_new, x_new) / DEG
    rlat = np.arcsin(z_new) / DEG
    return np.stack([rlat, rlon], axis=-1).astype(np.float32)


def rotated_to_wgs(rlats_deg, rlons_deg, sp_lat_deg, sp_lon_deg):
    """Inverse rotation (used by the synthetic-domain helpers and tests)."""
    r = rotation_constants(sp_lat_deg, sp_lon_deg)
    lat = np.asarray(rlats_deg, dtype=np.float64) * DEG
    lon = np.asarray(rlons_deg, dtype=np.float64) * DEG
    cl = np.cos(lat)
    x = np.cos(lon) * cl
    y = np.sin(lon) * cl
    z = np.sin(lat)
    # transpose of the forward rotation matrix
    x_o = r["ct"] * r["cp"] * x - r["sp"] * y - r["st"] * r["cp"] * z
    y_o = r["ct"] * r["sp"] * x + r["cp"] * y - r["st"] * r["sp"] * z
    z_o = r["st"] * x + r["ct"] * z
    return np.arcsin(z_o) / DEG, np.arctan2(y_o, x_o) / DEG
