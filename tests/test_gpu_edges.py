"""Edge cases of the HIP path against the oracle: sweeps without any hydrometeor, rays
that leave the model top or start below the topography, a one-gate sweep, a ragged set of
elevations in one call (cf. the reference's mask coding, interpolation.py:398-411, and
the all-NaN conventions of doppler_scatter.py:400-401, 472-477)."""
import copy

import numpy as np
import pytest

import _cases
from cosmo_pol_oracle import beam, scatter
from cosmo_pol_oracle import config as ocfg

pytestmark = pytest.mark.gpu

FIELDS = ['ZH', 'ZV', 'ZDR', 'RHOHV', 'KDP', 'ATT_H', 'ATT_V', 'DELTA_HV', 'PHIDP']


def _setup(name, over_extra=None, zero_hydro=False):
    from cosmo_pol_amd import RadarOperator
    over = copy.deepcopy(_cases.gen_golden.radial_case_inputs(name)[0])
    for sec, d in (over_extra or {}).items():
        over.setdefault(sec, {}).update(d)
    _, _, _, ocube, luts, cube = _cases.radial_case(name)
    if zero_hydro:
        cube = dict(cube, data={k: (np.zeros_like(v) if k.startswith('Q') else v)
                                for k, v in cube['data'].items()})
        for k in ocube.data:
            if k.startswith('Q'):
                ocube.data[k] = np.zeros_like(ocube.data[k])
    conf = ocfg.make_config(over)
    op = RadarOperator(config=over, luts=luts, output_variables='only_radar', lanes=1)
    op.load_model_arrays(cube['data'], cube['zlevels'], cube['proj_info'], cube['resolution'])
    return op, conf, ocube, {h: _cases.as_oracle_lut(l) for h, l in luts.items()}


def _compare(res, r, o, rtol=1e-5):
    for k in FIELDS:
        scale = np.nanmax(np.abs(o.values[k])) if np.isfinite(o.values[k]).any() else 0.0
        _cases.assert_close_nan(res[k][r], o.values[k], rtol=rtol, atol=1e-5 * scale, name=k)
    assert np.array_equal(res['mask'][r], o.mask)


def test_sweep_without_hydrometeors():
    """No valid (gate, hydrometeor) item at all: zero work units, every observable NaN."""
    op, conf, ocube, olut = _setup('c3_melt_ice', zero_hydro=True)
    azs, els = np.array([0., 120., 240.]), np.array([2., 5., 9.])
    res = op.simulate_rays(azs, els, apply_sensitivity=False)
    assert op._ctx.counters().n_valid_items == 0
    for r in range(3):
        subs = beam.interpolate_radial(ocube, conf, azs[r], els[r])
        o = scatter.radar_observables(subs, olut, conf)
        _compare(res, r, o)
        assert np.all(np.isnan(res['ZH'][r])) and np.all(np.isnan(res['RVEL'][r]))
    # and the same operator still works on the next, non-empty call
    op.close()


def test_rays_leaving_the_model_top_and_below_topography():
    """Steep rays exceed the model top (mask +1 -> sentinel -9999 -> NaN); a radar placed
    below the model topography starts under the lowest level (mask -1)."""
    op, conf, ocube, olut = _setup('c2_rsg', {'radar': {'range': 45000, 'radial_resolution': 500}})
    azs, els = np.array([30., 200., 310.]), np.array([35., 60., 89.])
    res = op.simulate_rays(azs, els, apply_sensitivity=False)
    n_top = 0
    for r in range(3):
        subs = beam.interpolate_radial(ocube, conf, azs[r], els[r])
        o = scatter.radar_observables(subs, olut, conf)
        _compare(res, r, o)
        n_top += int((o.mask == 1).sum())
    assert n_top > 10, 'no gate above the model top was exercised'
    op.close()

    low = {'radar': {'coords': [46.5, 7.5, -400.], 'range': 20000, 'radial_resolution': 250}}
    op, conf, ocube, olut = _setup('c2_rsg', low)
    azs, els = np.array([10., 100.]), np.array([0.5, -1.0])
    res = op.simulate_rays(azs, els, apply_sensitivity=False)
    n_below = 0
    for r in range(2):
        subs = beam.interpolate_radial(ocube, conf, azs[r], els[r])
        o = scatter.radar_observables(subs, olut, conf)
        _compare(res, r, o)
        n_below += int((o.mask == -1).sum())
    assert n_below > 5, 'no gate below the topography was exercised'
    op.close()


def test_one_gate_sweep_and_many_rays_of_mixed_elevation():
    # range == radial_resolution -> exactly one gate per ray
    op, conf, ocube, olut = _setup('c2_rsg', {'radar': {'range': 5000, 'radial_resolution': 5000}})
    assert len(op.constants.RANGE_RADAR) == 1
    azs, els = np.arange(0., 360., 45.), np.linspace(1., 20., 8)
    res = op.simulate_rays(azs, els, apply_sensitivity=False)
    assert res['ZH'].shape == (8, 1)
    for r in range(8):
        subs = beam.interpolate_radial(ocube, conf, azs[r], els[r])
        _compare(res, r, scatter.radar_observables(subs, olut, conf))
    op.close()


def test_bad_arguments_are_value_errors():
    op, conf, ocube, olut = _setup('c2_rsg')
    with pytest.raises(ValueError):
        op.simulate_rays([0., 1.], [1.])                  # ragged az / el
    with pytest.raises(ValueError):
        op.simulate_rays([0.], [1.], paths=np.zeros((1, 1, 3, 7), dtype=np.float32))
    op.close()
