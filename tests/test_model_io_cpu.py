"""Model files: NetCDF classic in COSMO's conventions and the .npz layout, read back into
what RadarOperator.load_model_arrays stages -- names, z-levels, proj_info and resolution as the reference
expects them of pycosmo (cosmo_pol/radar_operator.py:229-292, interpolation.py:547-561).  The files are
written by the test; GRIB is refused with a pointer."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cosmo_pol_amd import model_io, synthetic  # noqa: E402


def _raw_cube(two_mom=False):
    rng = np.random.default_rng(3)
    nz, ny, nx = 6, 5, 7
    hhl = np.linspace(12000., 300., nz + 1)[:, None, None] + rng.uniform(0, 200, (1, ny, nx))
    zf = 0.5 * (hhl[:-1] + hhl[1:])
    T = (288.0 - 6.5e-3 * zf).astype(np.float32)
    P = (101325.0 * np.exp(-zf / 8000.0)).astype(np.float32)
    raw = {'T': T, 'P': P, 'QV': np.full(T.shape, 3e-3, np.float32)}
    for k in ('QR', 'QC', 'QI', 'QS', 'QG'):
        raw[k] = rng.uniform(0, 1e-3, T.shape).astype(np.float32)
    for k in ('U', 'V'):
        raw[k] = rng.normal(0, 5, T.shape).astype(np.float32)
    raw['W'] = rng.normal(0, 1, (nz + 1, ny, nx)).astype(np.float32)         # half levels, as COSMO writes it
    if two_mom:
        raw['QH'] = rng.uniform(0, 1e-4, T.shape).astype(np.float32)
        for k in ('QNH', 'QNR', 'QNS', 'QNG'):
            raw[k] = rng.uniform(1, 1e4, T.shape).astype(np.float32)
    rlon = -1.0 + 0.02 * np.arange(nx)
    rlat = 0.5 + 0.02 * np.arange(ny)
    return raw, hhl.astype(np.float32), rlon, rlat


@pytest.mark.parametrize('two_mom', [False, True])
def test_netcdf_classic_raw_output_with_cfile(tmp_path, two_mom):
    raw, hhl, rlon, rlat = _raw_cube(two_mom)
    f, c = str(tmp_path / 'lfff00000000.nc'), str(tmp_path / 'lfff00000000c.nc')
    model_io.write_netcdf(f, raw, rlon, rlat, north_pole=(43.0, -170.0))
    model_io.write_netcdf(c, {'FR_LAND': np.zeros((1,) + hhl.shape[1:], np.float32)[:1].repeat(hhl.shape[0] - 1, 0)},
                          rlon, rlat, north_pole=(43.0, -170.0), hhl=hhl)
    m = model_io.read_model_file(f, c, want_refractivity=True)
    assert m['scheme'] == ('2mom' if two_mom else '1mom') and m['derived_from_raw']
    want = set(model_io.BASE_VARIABLES) | {'N'} | (set(model_io.BASE_VARIABLES_2MOM) if two_mom else set())
    assert set(m['data']) == want
    nz = raw['T'].shape[0]
    for k, v in m['data'].items():
        assert v.dtype == np.float32 and v.shape == raw['T'].shape, k
    assert np.allclose(m['zlevels'], 0.5 * (hhl[:-1] + hhl[1:])) and m['zlevels'][0].mean() > m['zlevels'][-1].mean()
    # densities: mass ratio x air density of moist air with its condensate
    T, P, QV = (raw[k].astype(np.float64) for k in ('T', 'P', 'QV'))
    load = sum(raw[k].astype(np.float64) for k in ('QC', 'QR', 'QS', 'QG', 'QI'))
    rho = P / (287.05 * T * (1 + (461.51 / 287.05 - 1) * QV - load))
    assert np.allclose(m['data']['RHO'], rho, rtol=1e-6) and 0.2 < rho.min() and rho.max() < 1.4
    assert np.allclose(m['data']['QR_v'], raw['QR'] * rho, rtol=1e-6)
    assert np.allclose(m['data']['W'], 0.5 * (raw['W'][:-1] + raw['W'][1:]), atol=1e-6)
    assert 50 < m['data']['N'].min() and m['data']['N'].max() < 450                  # N-units
    if two_mom:
        assert np.allclose(m['data']['QNR_v'], raw['QNR'] * rho, rtol=1e-6)
        assert not m['data']['QNI_v'].any()                                           # not in the file: zeros
    pi = m['proj_info']
    assert pi['Latitude_of_southern_pole'] == -43.0 and pi['Longitude_of_southern_pole'] == 10.0
    assert pi['Lo1'] == rlon[0] and pi['La2'] == rlat[-1]
    assert np.allclose(m['resolution'], (0.02, 0.02))
    assert 'hours since' in str(m['time'])


def test_npz_with_derived_variables_round_trip(tmp_path):
    cube = synthetic.small_test_cube(hydrometeors=('R', 'S', 'G'))
    f = str(tmp_path / 'cube.npz')
    model_io.write_npz(f, cube['data'], zlevels=cube['zlevels'], proj_info=cube['proj_info'], time='2014-08-13 12:00')
    m = model_io.read_model_file(f)
    assert m['scheme'] == '1mom' and not m['derived_from_raw']
    for k in model_io.BASE_VARIABLES:
        assert np.array_equal(m['data'][k], cube['data'][k]), k
    assert np.array_equal(m['zlevels'], cube['zlevels'])
    assert m['proj_info'] == {k: float(cube['proj_info'][k]) for k in model_io.PROJ_KEYS}
    assert np.allclose(m['resolution'], cube['resolution'])
    assert m['time'] == '2014-08-13 12:00'


def test_missing_variable_raises_like_the_reference_and_grib_is_refused(tmp_path):
    raw, hhl, rlon, rlat = _raw_cube()
    del raw['QG']
    f = str(tmp_path / 'incomplete.npz')
    model_io.write_npz(f, raw, hhl=hhl, rlon=rlon, rlat=rlat, north_pole=(43.0, -170.0))
    with pytest.raises(ValueError, match='Not all necessary variables'):
        model_io.read_model_file(f)
    raw, hhl, rlon, rlat = _raw_cube()
    f2 = str(tmp_path / 'noheights.npz')
    model_io.write_npz(f2, raw, rlon=rlon, rlat=rlat, north_pole=(43.0, -170.0))
    with pytest.raises(ValueError, match='no level heights'):
        model_io.read_model_file(f2)
    g = str(tmp_path / 'laf2014081312.grb')
    with open(g, 'wb') as fh:
        fh.write(b'GRIB' + b'\x00' * 64)
    with pytest.raises(NotImplementedError, match='pycosmo'):
        model_io.read_model_file(g)
