"""CPU tests of the RadarScan / SimulatedGPM containers (PyartRadop and
SimulatedGPM conventions, cosmo_pol/radar/pyart_wrapper.py:186-342,
cosmo_pol/radar/gpm_wrapper.py:47-141)."""
import numpy as np

from cosmo_pol_amd import gpm
from cosmo_pol_amd.radar_operator import RadarScan


def _sweep(n_rays, n_gates, el, seed):
    rng = np.random.default_rng(seed)
    f = {k: rng.uniform(0.5, 50, (n_rays, n_gates)).astype(np.float32) for k in ('ZH', 'ZV', 'ZDR', 'KDP')}
    f['ZH'][0, :3] = np.nan
    f['ZH'][1, 4] = 0.0
    az = np.arange(n_rays, dtype=float) * 10
    return {'fields': f, 'azimuth': az, 'elevation': np.full(n_rays, el),
            'lats': rng.normal(size=(n_rays, n_gates)), 'lons': rng.normal(size=(n_rays, n_gates)),
            'mask': np.zeros((n_rays, n_gates)), 'dist': None, 'heights': None}


def test_radar_scan_conventions():
    sw = [_sweep(6, 9, 1.0, 1), _sweep(6, 9, 2.5, 2)]
    rr = np.arange(9) * 300.0 + 150
    scan = RadarScan('ppi', [1.0, 2.5], list(sw[0]['azimuth']), rr,
                     {'latitude': 46.5, 'longitude': 7.5, 'altitude': 1000, 'time': None}, sw)
    assert scan.nsweeps == 2 and scan.nrays == 12 and scan.ngates == 9
    zh = scan.fields['ZH']['data']
    assert zh.shape == (12, 9) and zh.mask[0, :3].all() and zh.mask[1, 4]     # NaN and 0 masked
    with np.errstate(divide='ignore', invalid='ignore'):
        exp = 10 * np.log10(sw[1]['fields']['ZH'])
    exp[~np.isfinite(exp)] = np.nan                 # 0 -> NaN before the dB conversion
    assert np.allclose(scan.get_field(1, 'ZH').filled(np.nan), exp, equal_nan=True)
    assert np.array_equal(scan.fields['KDP']['data'][:6], sw[0]['fields']['KDP'])   # not in dB
    assert np.array_equal(scan.fields['rangearray']['data'][5], rr)
    assert np.array_equal(scan.sweep_start_ray_index['data'], [0, 6])
    assert np.array_equal(scan.sweep_stop_ray_index['data'], [5, 11])
    assert np.array_equal(scan.elevation['data'], [1.0] * 6 + [2.5] * 6)
    assert 'Latitude' in scan.fields and 'Longitude' in scan.fields


def test_simulated_gpm_packaging():
    n_rays, n_gates = 6, 10
    mask = np.zeros((n_rays, n_gates))
    mask[:, :2] = 1            # above the model top
    mask[:, 8:] = -1           # under the topography
    n_kept = np.array([10, 10, 9, 10, 8, 10])
    f = {'ZH': np.arange(n_rays * n_gates, dtype=float).reshape(n_rays, n_gates)}
    lats = f['ZH'] + 0.5
    out = gpm.SimulatedGPM(f, mask, lats, lats + 1, n_kept, (2, 3), 'Ku')
    assert out.data['ZH'].shape == (2, 3, n_gates) and out.band == 'Ku'
    # beam 0: gates with mask > -1 are 0..7, flipped so that index 0 is nearest the ground
    assert np.array_equal(out.data['ZH'][0, 0, :8], f['ZH'][0, :8][::-1])
    assert np.all(out.data['ZH'][0, 0, 8:] == 0) and np.isnan(out.lats[0, 0, 8:]).all()
    assert out.bin_surface[0, 0] == 10 and out.bin_surface[0, 2] == 9
    # beam 4 has only 8 gates (all above ground except none below) -> 8 values
    assert np.array_equal(out.data['ZH'][1, 1, :8], f['ZH'][4, :8][::-1])


def test_gpm_geometry_sanity():
    sw = gpm.synthetic_swath(n_scans=4, n_rays=7)
    az, el, rng, sat = gpm.swath_angles(sw)
    assert az.shape == (4, 7) and np.all(el > 70) and np.all(el <= 90)
    assert abs(el[0, 3] - 90) < 1e-4 and abs(rng[0, 3] - 407000) < 1.0          # nadir
    assert np.all(np.diff(rng[0, 3:]) > 0)                                        # slant range grows off-nadir
    assert gpm.band_settings('Ka') == (35.6, 250) and gpm.band_settings('Ku_matched') == (13.6, 125)
    # inverse azimuth against the direct problem of the oracle
    from cosmo_pol_oracle import geodesy
    lat2, lon2 = geodesy.wgs84_direct(46.5, 7.5, 63.0, np.array([80000.0]))
    assert abs(gpm.wgs84_inverse_azimuth(46.5, 7.5, lat2[0], lon2[0]) - 63.0) < 1e-9


def test_simulated_gpm_packing_matches_per_ray_definition():
    """SimulatedGPM (gpm_wrapper.py:47-141): per ray, drop the gates below the topography,
    flip the beam so that index 0 is the lowest gate, left-align; bin_surface counts from the
    first gate above the model top.  Vectorised packing against the per-ray definition."""
    import numpy as np
    from cosmo_pol_amd.gpm import SimulatedGPM
    rng = np.random.default_rng(1)
    N, M, G = 4, 5, 30
    n_kept = rng.integers(0, G + 1, N * M)
    mask = rng.choice([-1., 0., 0., 1., 0.3], size=(N * M, G))
    lats = rng.random((N * M, G))
    lons = rng.random((N * M, G))
    f = {'ZH': rng.random((N * M, G)).astype(np.float32)}
    o = SimulatedGPM(f, mask, lats, lons, n_kept, (N, M), 'Ku')
    for idx in range(N * M):
        i, j = divmod(idx, M)
        L = int(n_kept[idx])
        m = mask[idx, :L]
        above = np.where(m >= 1)[0]
        assert o.bin_surface[i, j] == ((L - above[0]) if len(above) else 0)
        keep = m > -1
        n = int(keep.sum())
        exp = np.zeros(G)
        exp[:n] = f['ZH'][idx, :L][keep][::-1]
        assert np.array_equal(o.data['ZH'][i, j], exp)
        el = np.full(G, np.nan)
        el[:n] = lats[idx, :L][keep][::-1]
        assert np.array_equal(o.lats[i, j], el, equal_nan=True)


def test_pyart_packaging_passes_the_reference_arguments(monkeypatch):
    """as_pyart_radar: with a (stand-in) pyart.core.Radar present, the scan is handed over with
    the positional arguments of the reference's PyartRadop (pyart_wrapper.py:335-339), the
    reference's field metadata and the Doppler velocity bins."""
    import sys
    import types
    from cosmo_pol_amd import pyart_wrapper as PW
    from cosmo_pol_amd.radar_operator import RadarScan

    class FakeRadar(object):
        def __init__(self, time, _range, fields, metadata, scan_type, latitude, longitude, altitude,
                     sweep_number, sweep_mode, fixed_angle, sweep_start_ray_index,
                     sweep_stop_ray_index, azimuth, elevation, instrument_parameters=None):
            self.__dict__.update(locals())
            self.range = _range

    pyart = types.ModuleType('pyart')
    core = types.ModuleType('pyart.core')
    core.Radar = FakeRadar
    pyart.core = core
    monkeypatch.setitem(sys.modules, 'pyart', pyart)
    monkeypatch.setitem(sys.modules, 'pyart.core', core)
    assert PW.pyart_available()
    n_g = 6
    rng = np.random.default_rng(0)

    def sweep(n_r, el):
        f = {k: (10 ** rng.uniform(-1, 3, (n_r, n_g))).astype(np.float32) for k in ('ZH', 'ZDR', 'KDP')}
        f['ZH'][0, 0] = 0.0
        f['KDP'][1, 2] = np.nan
        return {'fields': f, 'azimuth': np.arange(n_r, dtype=float), 'elevation': np.full(n_r, el),
                'lats': rng.normal(size=(n_r, n_g)), 'lons': rng.normal(size=(n_r, n_g)),
                'mask': np.zeros((n_r, n_g)), 'dist': np.zeros((n_r, n_g), dtype=np.float32),
                'heights': np.zeros((n_r, n_g), dtype=np.float32)}
    scan = RadarScan('ppi', [1.0, 2.0], list(np.arange(3.)), np.arange(n_g) * 300.,
                     {'latitude': 46.5, 'longitude': 7.5, 'altitude': 1000., 'time': '2026-03-01'},
                     [sweep(3, 1.0), sweep(3, 2.0)])
    radar = scan.to_pyart(varray=np.linspace(-8, 8, 5))
    assert isinstance(radar, FakeRadar) and radar.scan_type == 'ppi'
    assert radar.fields['ZH']['units'] == 'dBZ' and radar.fields['KDP']['long_name'] == 'Specific diff. phase'
    assert radar.fields['ZH']['valid_max'] == 55 and radar.fields['Latitude']['units'] == ['degrees']
    assert radar.fields['ZH']['data'].shape == (6, n_g) and np.ma.is_masked(radar.fields['ZH']['data'][0, 0])
    assert np.array_equal(radar.sweep_start_ray_index['data'], [0, 3])
    assert np.allclose(radar.fixed_angle['data'], [1.0, 2.0])
    assert np.array_equal(radar.instrument_parameters['varray']['data'], np.linspace(-8, 8, 5))
    assert radar.get_field(1, 'ZDR').shape == (3, n_g)
    assert radar.fields['rangearray']['data'].shape == (6, n_g)


def test_lazy_fields_equal_the_eager_container():
    """RadarScan.fields builds a field (stack over the sweeps, dB with 0 -> NaN, NaN mask) when it is
    first read: same arrays as the eager construction, every dict access path goes through the
    builder, untouched fields cost nothing."""
    from cosmo_pol_amd.radar_operator import LazyDict, RadarScan
    rng = np.random.default_rng(3)
    n_g = 7
    sweeps = []
    for n in (4, 3):
        f = {k: rng.uniform(0.1, 5.0, size=(n, n_g)).astype(np.float32) for k in ('ZH', 'ZDR', 'ZV', 'KDP')}
        f['ZH'][0, 1] = 0.0
        f['KDP'][1, 2] = np.nan
        sweeps.append({'fields': f, 'azimuth': np.arange(n, dtype=float), 'elevation': np.full(n, 1.0),
                       'lats': rng.normal(size=(n, n_g)), 'lons': rng.normal(size=(n, n_g))})
    scan = RadarScan('ppi', [1.0, 2.0], [0., 1., 2., 3.], np.arange(n_g) * 100., dict(latitude=46., longitude=7., altitude=500., time=None), sweeps)
    assert isinstance(scan.fields, LazyDict)
    assert list(scan.fields.keys()) == ['ZH', 'ZDR', 'ZV', 'KDP', 'Latitude', 'Longitude', 'rangearray']
    assert 'ZH' in scan.fields and len(scan.fields) == 7 and 'pending' in repr(scan.fields)
    assert scan.fields.pending('ZH')                                # nothing built yet
    zh = scan.fields['ZH']['data']
    assert not scan.fields.pending('ZH') and scan.fields.pending('ZV')
    lin = np.concatenate([s['fields']['ZH'] for s in sweeps])
    with np.errstate(divide='ignore'):
        exp = 10 * np.log10(np.where(lin == 0, np.nan, lin))
    assert zh.shape == (7, n_g) and zh.mask[0, 1] and np.allclose(zh.filled(np.nan), exp, equal_nan=True)
    assert sweeps[0]['fields']['ZH'][0, 1] == 0.0                   # the raw (linear) arrays are untouched
    kdp = scan.fields.get('KDP')['data']
    assert kdp.mask[1, 2] and kdp.mask[5, 2] and kdp.mask.sum() == 2
    assert scan.get_field(1, 'ZV').shape == (3, n_g)
    for k, v in scan.fields.items():
        assert v['data'].shape == (7, n_g), k
    assert scan.fields['rangearray']['data'][3, 2] == 200.
    assert [int(x) for x in scan.sweep_start_ray_index['data']] == [0, 4]
    assert [int(x) for x in scan.sweep_stop_ray_index['data']] == [3, 6]
    assert scan.nrays == 7 and len(scan.azimuth['data']) == 7
    scan.fields['extra'] = {'data': 1}
    assert list(scan.fields)[-1] == 'extra'
    del scan.fields['ZDR']
    assert 'ZDR' not in scan.fields and len(scan.fields) == 7


def test_lazydict_is_a_full_mapping():
    import pytest
    """Round-3 advisor finding: the containers' field mapping must behave like the plain dict of the
    reference for every dict method, pending entries included."""
    import copy
    import pickle
    from cosmo_pol_amd.radar_operator import LazyDict
    built = []

    def mk(v):
        def make():
            built.append(v)
            return v
        return make
    d = LazyDict()
    d.add('a', mk(1))
    d.add('b', mk(2))
    d['c'] = 3
    assert list(d) == ['a', 'b', 'c'] and len(d) == 3 and 'a' in d and not built
    assert d.pop('a') == 1 and built == [1] and list(d) == ['b', 'c']          # pop of a pending key builds it
    d.update({'x': 9, 'b': 20})                                                # update sees and orders every key
    assert list(d.keys()) == ['b', 'c', 'x'] and d['b'] == 20 and built == [1]  # (the overwritten builder never ran)
    assert d.setdefault('c', 0) == 3 and d.setdefault('y', 7) == 7
    d.add('z', mk(5))
    assert d == {'b': 20, 'c': 3, 'x': 9, 'y': 7, 'z': 5}                      # == builds and compares values
    c = d.copy()
    assert type(c) is dict and c == dict(d.items())
    d.add('p', mk(11))
    assert pickle.loads(pickle.dumps(d)) == dict(d.items()) and copy.deepcopy(d)['p'] == 11
    del d['p']
    with pytest.raises(KeyError):
        d['p']
    assert d.get('nope', 4) == 4 and 'pending' not in repr(d)
