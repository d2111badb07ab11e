"""cosmo_pol_amd.radial: batched results as the reference's per-radial records
(cosmo_pol/interpolation/radial.py:17-54).  CPU part: structure, views, and the oracle's
cut_at_sensitivity running on the records.  (GPU part: tests/test_gpu_boundary.py.)"""
import numpy as np

from cosmo_pol_amd import radial
from cosmo_pol_oracle import config as ocfg
from cosmo_pol_oracle import scatter


def _fake_result(n_rays=3, n_gates=7, seed=0):
    rng = np.random.default_rng(seed)
    res = {k: (10 ** rng.uniform(-3, 3, (n_rays, n_gates))).astype(np.float32)
           for k in ['ZH', 'ZV', 'ZDR', 'KDP', 'DELTA_HV', 'PHIDP', 'RHOHV', 'ATT_H', 'ATT_V']}
    res['RVEL'] = rng.normal(size=(n_rays, n_gates))
    res['mask'] = np.zeros((n_rays, n_gates))
    res['lats'] = rng.normal(size=(n_rays, n_gates))
    res['lons'] = rng.normal(size=(n_rays, n_gates))
    res['dist'] = np.tile(np.arange(n_gates, dtype=np.float32) * 300, (n_rays, 1))
    res['heights'] = res['dist'] * 0.02
    res['n_sub'] = 1
    return res


def test_radial_has_the_reference_attributes():
    r = radial.Radial({'ZH': np.ones(4)}, np.zeros(4), np.zeros(4), np.zeros(4), np.zeros(4), np.zeros(4))
    for a in ('mask', 'quad_pt', 'quad_weight', 'lats_profile', 'lons_profile', 'dist_profile',
              'heights_profile', 'elev_profile', 'values', 'has_melting', 'mask_ml'):
        assert hasattr(r, a), a
    assert r.quad_weight == 1 and r.has_melting is False and r.mask_ml is None


def test_to_radials_rows_are_views_and_cut_edits_the_batch():
    res = _fake_result()
    rads = radial.to_radials(res, azimuths=[0., 1., 2.], elevations=[1., 1., 1.])
    assert len(rads) == 3 and rads[1].quad_pt == [1.0, 1.0]
    assert set(rads[0].values) == {'ZH', 'ZV', 'ZDR', 'KDP', 'DELTA_HV', 'PHIDP', 'RHOHV', 'ATT_H',
                                   'ATT_V', 'RVEL'}
    assert np.shares_memory(rads[2].values['ZH'], res['ZH'])
    assert np.array_equal(rads[1].dist_profile, res['dist'][1])
    conf = ocfg.make_config({'radar': {'coords': [46.5, 7.5, 1000], 'frequency': 5.6, 'range': 2100,
                                       'radial_resolution': 300, 'sensitivity': [20., 10000]}})
    before = {k: res[k].copy() for k in ('ZH', 'KDP', 'ATT_H')}
    scatter.cut_at_sensitivity([rads], conf)                 # list of lists, as get_PPI does
    thr = scatter.sensitivity_threshold(conf, 7)
    with np.errstate(divide='ignore'):
        m = 10 * np.log10(before['ZH']) < thr[None]
    assert m.any() and (~m).any()
    assert np.array_equal(np.isnan(res['ZH']), m) and np.array_equal(np.isnan(res['KDP']), m)
    assert np.array_equal(res['ATT_H'], before['ATT_H'])     # not a simulated variable: untouched


def test_packaged_sweep_form():
    res = _fake_result()
    sweep = {'fields': {k: res[k] for k in ('ZH', 'ZDR')}, 'azimuth': np.arange(3.), 'elevation': np.ones(3),
             'mask': res['mask'], 'lats': res['lats'], 'lons': res['lons'], 'dist': res['dist'],
             'heights': res['heights']}
    rads = radial.to_radials(sweep)
    assert set(rads[0].values) == {'ZH', 'ZDR'} and rads[2].quad_pt == [2.0, 1.0]
