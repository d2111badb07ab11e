"""Host side of the antenna quadratures (cosmo_pol_amd/quadrature.py) against the
reference's own sub-beam lists (tests/golden/radial_q_*.npz: quad_pts / quad_w written
by get_interpolated_radial, interpolation.py:148-256, 342-354, 423-436)."""
import numpy as np
import pytest

import _cases
from cosmo_pol_amd import config as cfg
from cosmo_pol_amd import quadrature

CASES = ['q_ml', 'q_ml_thr', 'q_legendre', 'q_multigauss']


def _config(name):
    return cfg.sanity_check(_cases.gen_golden.radial_case_inputs(name)[0])


@pytest.mark.parametrize('name', CASES)
def test_subbeams_match_reference(golden, name):
    g = golden('radial_' + name)
    sb = quadrature.subbeams(_config(name))
    assert sb.n_sub == int(g['n_sub'])
    pts = np.stack([sb.pts_hor[sb.sub_h] + float(g['azimuth']),
                    sb.pts_ver[sb.sub_v] + float(g['elevation'])], axis=1)
    np.testing.assert_allclose(pts, g['quad_pts'], rtol=0, atol=1e-12)
    qw = g['quad_w']
    if sb.sub_smooth is None:
        np.testing.assert_allclose(sb.sub_w, qw[:, 0], rtol=1e-13)
        assert np.all(qw == qw[:, :1])
    else:
        plain = sb.sub_smooth == 0
        np.testing.assert_allclose(sb.sub_w[plain], qw[plain, 0], rtol=1e-13)
        assert np.all(qw[plain] == qw[plain][:, :1])
        # edge-mask sub-beams: w x (sum of at most two shifted filter kernels)
        r, taps = quadrature.ml_filter_taps()
        assert r == 8 and len(taps) == 17 and abs(taps.sum() - 1) < 1e-15
        for s in np.where(~plain)[0]:
            ratio = qw[s] / sb.sub_w[s]
            assert ratio.max() <= 2 * taps.max() + 1e-15
            if ratio.max() > 0:
                assert abs(ratio.sum() - 2.0) < 1e-12 or abs(ratio.sum() - 1.0) < 1e-12


def test_filter_taps_are_scipys():
    from scipy.ndimage import gaussian_filter
    r, taps = quadrature.ml_filter_taps()
    x = np.zeros(41)
    x[20] = 1
    assert np.array_equal(gaussian_filter(x, 2)[20 - r:20 + r + 1], taps)


def test_point_list_layout():
    sb = quadrature.subbeams(_config('q_multigauss'))
    assert np.array_equal(sb.sub_h, np.arange(sb.n_sub)) and np.array_equal(sb.sub_v, sb.sub_h)
    assert len(sb.pts_hor) == len(sb.pts_ver) == sb.n_sub


def test_antenna_fit_and_rejected_schemes():
    over = _cases.gen_golden.radial_case_inputs('q_multigauss')[0]
    over = {k: dict(v) for k, v in over.items()}
    over['integration'].pop('antenna_params')
    over['integration'].update(antenna_diagram=_cases.gen_golden.ANTENNA_CSV, n_gaussians=3)
    c = cfg.sanity_check(over)
    ap = np.asarray(c['integration']['antenna_params'])
    assert ap.shape == (3, 3) and ap[0, 0] == 0 and ap[0, 1] == 0 and 0.1 <= ap[0, 2] <= 2
    assert quadrature.subbeams(c).n_sub > 0
    c['integration']['scheme'] = 4
    with pytest.raises(NotImplementedError):
        quadrature.subbeams(c)
    # schemes the reference's VALID_VALUES reject fall back to the default with a notice
    over['integration']['scheme'] = 6
    assert cfg.sanity_check(over)['integration']['scheme'] == 1
