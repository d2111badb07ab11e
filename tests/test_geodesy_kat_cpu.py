"""Known-answer tests of the geodesy that stands in for pyproj.Geod.fwd and
pycosmo.WGS_to_COSMO (reference call sites interpolation/interpolation.py:526-536, 566-568;
the packages' sources are absent): these tests are the only evidence for
that boundary, everything downstream is pinned by the reference itself.

  * Vincenty direct vs the Flinders Peak -> Buninyong line (Geoscience Australia's worked
    example of Vincenty's formulae: 54 972.271 m at azimuth 306 52' 05.37");
  * vs an independent numerical integration of the geodesic ODE on the WGS84 ellipsoid;
  * Clairaut's constant along the line; antipodal / meridian / equator special cases;
  * rotated pole vs the COSMO model's own phi2phirot / rla2rlarot formulas (spherical
    trigonometry instead of the rotation matrix used here), pole / origin known answers and
    the round trip.
The HIP kernel mirrors oracle/cosmo_pol_oracle/geodesy.py operation by operation
(tests/test_gpu_parity.py compares rotated coordinates bit for bit).
"""
import numpy as np
import pytest
from scipy.integrate import solve_ivp

from cosmo_pol_oracle import geodesy as G


def _dms(d, m, s):
    sign = -1.0 if d < 0 else 1.0
    return sign * (abs(d) + m / 60.0 + s / 3600.0)


def test_vincenty_direct_flinders_peak_to_buninyong():
    # GDA technical manual, "Vincenty's formulae" worked example (GRS80; WGS84 differs from
    # GRS80 by 0.1 mm in the semi-minor axis: < 1e-9 deg over this line)
    lat1, lon1 = _dms(-37, 57, 3.72030), _dms(144, 25, 29.52440)
    lat2, lon2 = _dms(-37, 39, 10.15610), _dms(143, 55, 35.38390)
    az12 = _dms(306, 52, 5.37)
    s = 54972.271
    lat, lon = G.wgs84_direct(lat1, lon1, az12, np.array([s]))
    # the published azimuth is rounded to 0.01" (2.8e-6 deg -> 2.7 mm across the line at
    # 55 km = 3e-8 deg); the distance to 1 mm (1e-8 deg)
    assert abs(lat[0] - lat2) < 4e-8, lat[0] - lat2
    assert abs(lon[0] - lon2) < 4e-8, lon[0] - lon2


def _geodesic_ode(lat1, lon1, az, s_end):
    """Independent solution: d(phi, lambda, alpha)/ds on the ellipsoid, DOP853."""
    a, f = G.WGS84_A, G.WGS84_F
    e2 = f * (2.0 - f)

    def rhs(_, y):
        phi, lam, alpha = y
        w = np.sqrt(1.0 - e2 * np.sin(phi) ** 2)
        M = a * (1.0 - e2) / w ** 3
        Nn = a / w
        return [np.cos(alpha) / M, np.sin(alpha) / (Nn * np.cos(phi)),
                np.sin(alpha) * np.tan(phi) / Nn]
    sol = solve_ivp(rhs, (0.0, s_end), [np.deg2rad(lat1), np.deg2rad(lon1), np.deg2rad(az)],
                    method='DOP853', rtol=1e-13, atol=1e-15)
    return np.rad2deg(sol.y[0, -1]), np.rad2deg(sol.y[1, -1]), np.rad2deg(sol.y[2, -1])


@pytest.mark.parametrize('site', [(46.5, 7.5), (-37.95, 144.42), (0.3, -75.0), (69.0, 18.0)])
def test_vincenty_direct_vs_geodesic_ode(site):
    lat1, lon1 = site
    worst = 0.0
    for az in (0.0, 37.0, 90.0, 143.5, 180.0, 222.0, 270.0, 359.0):
        for s in (150.0, 15e3, 150e3, 407e3, 1200e3):      # radar gates ... GPM slant ranges
            lat, lon = G.wgs84_direct(lat1, lon1, az, np.array([s]))
            elat, elon, _ = _geodesic_ode(lat1, lon1, az, s)
            dlon = (lon[0] - elon + 180.0) % 360.0 - 180.0
            worst = max(worst, abs(lat[0] - elat), abs(dlon) * np.cos(np.deg2rad(elat)))
    # 5 fixed Vincenty iterations vs the ODE: below 1e-10 deg (0.01 mm)
    assert worst < 1e-10, worst


def test_clairaut_constant_and_special_lines():
    a, f = G.WGS84_A, G.WGS84_F
    lat1, lon1, az = 46.5, 7.5, 63.0
    s = np.linspace(0.0, 400e3, 9)
    lat, lon = G.wgs84_direct(lat1, lon1, az, s)
    # Clairaut: cos(U) sin(alpha) constant; alpha from finite differences of the line itself
    k = G.direct_ray_constants(lat1, lon1, az)
    U = np.arctan((1 - f) * np.tan(np.deg2rad(lat)))
    ds = 1.0
    la, lo = G.wgs84_direct(lat1, lon1, az, s + ds)
    e2 = f * (2 - f)
    w = np.sqrt(1 - e2 * np.sin(np.deg2rad(lat)) ** 2)
    north = np.deg2rad(la - lat) * a * (1 - e2) / w ** 3
    east = np.deg2rad(lo - lon) * a / w * np.cos(np.deg2rad(lat))
    alpha = np.arctan2(east, north)
    np.testing.assert_allclose(np.cos(U) * np.sin(alpha), k['sin_alpha'], rtol=0, atol=2e-7)
    # due north along a meridian: longitude unchanged, meridian arc length of WGS84
    lat_n, lon_n = G.wgs84_direct(0.0, 10.0, 0.0, np.array([110574.38855779878]))
    assert abs(lon_n[0] - 10.0) < 1e-12
    assert abs(lat_n[0] - 1.0) < 1e-9          # one degree of meridian at the equator: 110 574.389 m
    # due east along the equator: latitude stays 0, lon = s / a
    lat_e, lon_e = G.wgs84_direct(0.0, 10.0, 90.0, np.array([250e3]))
    assert abs(lat_e[0]) < 1e-12
    assert abs(lon_e[0] - (10.0 + np.rad2deg(250e3 / a))) < 1e-10


def _cosmo_phi2phirot(phi, rla, pollat, pollon):
    """COSMO model utilities phi2phirot / rla2rlarot (polgam = 0), as documented in the COSMO
    model documentation (spherical trigonometry with the rotated NORTH pole)."""
    zsinpol, zcospol = np.sin(np.deg2rad(pollat)), np.cos(np.deg2rad(pollat))
    zlampol = np.deg2rad(pollon)
    zphi, zrla = np.deg2rad(phi), np.deg2rad(np.where(rla > 180.0, rla - 360.0, rla))
    zarg = zcospol * np.cos(zphi) * np.cos(zrla - zlampol) + zsinpol * np.sin(zphi)
    phirot = np.rad2deg(np.arcsin(zarg))
    zarg1 = -np.sin(zrla - zlampol) * np.cos(zphi)
    zarg2 = -zsinpol * np.cos(zphi) * np.cos(zrla - zlampol) + zcospol * np.sin(zphi)
    rlarot = np.rad2deg(np.arctan2(zarg1, zarg2))
    return phirot, rlarot


@pytest.mark.parametrize('south_pole', [(-43.0, 10.0), (-40.0, 10.0), (-32.5, -170.0)])
def test_rotated_pole_vs_cosmo_formulas_and_round_trip(south_pole):
    sp_lat, sp_lon = south_pole
    pollat, pollon = -sp_lat, sp_lon - 180.0 if sp_lon > 0 else sp_lon + 180.0
    rng = np.random.default_rng(4)
    lat = rng.uniform(-80, 85, 400)
    lon = rng.uniform(-179, 179, 400)
    rc = G.wgs_to_rotated(lat, lon, sp_lat, sp_lon)
    assert rc.dtype == np.float32 and rc.shape == (400, 2)          # contract of the call sites
    prot, lrot = _cosmo_phi2phirot(lat, lon, pollat, pollon)
    # float32 output: compare at float32 resolution of angles up to 180 deg
    np.testing.assert_allclose(rc[:, 0], prot, rtol=0, atol=8e-6)
    dl = (rc[:, 1].astype(np.float64) - lrot + 180.0) % 360.0 - 180.0
    assert np.max(np.abs(dl) * np.cos(np.deg2rad(prot))) < 2e-5
    # known answers: the rotated north pole, the rotated origin
    np.testing.assert_allclose(G.wgs_to_rotated([pollat], [pollon], sp_lat, sp_lon)[0, 0], 90.0, atol=1e-4)
    origin = G.wgs_to_rotated([90.0 + sp_lat], [sp_lon], sp_lat, sp_lon)[0]
    np.testing.assert_allclose(origin, [0.0, 0.0], atol=1e-5)
    np.testing.assert_allclose(G.wgs_to_rotated([sp_lat], [sp_lon], sp_lat, sp_lon)[0, 0], -90.0, atol=1e-4)
    # round trip through the inverse rotation (float32 rotated coordinates in between)
    la, lo = G.rotated_to_wgs(rc[:, 0], rc[:, 1], sp_lat, sp_lon)
    np.testing.assert_allclose(la, lat, rtol=0, atol=2e-5)
    dlo = (lo - lon + 180.0) % 360.0 - 180.0
    assert np.max(np.abs(dlo) * np.cos(np.deg2rad(lat))) < 2e-5


def test_product_host_constants_equal_the_oracle():
    """cosmo_pol_amd.geometry (the host half of the device geodesic) evaluates the same
    per-ray constants as the oracle, bit for bit."""
    from cosmo_pol_amd import geometry as geo
    coords = [46.5, 7.5, 1000.0]
    sin_u1, cos_u1, _ = geo.radar_site_constants(coords)
    k = G.direct_ray_constants(coords[0], coords[1], np.array([0.0, 33.0, 271.5]))
    assert sin_u1 == k['sin_u1'] and cos_u1 == k['cos_u1']
