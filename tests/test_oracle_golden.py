"""CPU tests: the oracle (oracle/cosmo_pol_oracle + oracle/interp_twin.c) against
the golden vectors produced by the reference itself (oracle/gen_golden.py)."""
import numpy as np
import pytest

import _cases
from cosmo_pol_oracle import beam, psd, scatter
from cosmo_pol_oracle import config as ocfg
from cosmo_pol_oracle import constants as OK


def test_gate_kernel_twin_bit_exact(golden):
    g = golden('gate_kernel')
    out = beam.get_all_radar_pts(g['coords'], g['heights'], g['data'], g['zlevels'], g['llc'],
                                 g['res'], which='twin')
    exp = g['expected']
    assert np.array_equal(np.isnan(out), np.isnan(exp))
    ok = ~np.isnan(exp)
    assert np.array_equal(out[ok].view(np.uint32), exp[ok].view(np.uint32))
    # sentinel / mask classes all present in the fixture
    assert (exp == -9999).sum() > 5 and np.isnan(exp).sum() > 5 and ok.sum() > 100


def test_binary_search_known_answers(golden):
    import ctypes
    g = golden('gate_kernel')
    lib = beam._load_interp_lib('twin')
    lib.binary_search.restype = ctypes.c_int
    lib.binary_search.argtypes = [ctypes.POINTER(ctypes.c_float), ctypes.c_int, ctypes.c_float]
    arr = np.ascontiguousarray(g['bs_arr'])
    got = [lib.binary_search(arr.ctypes.data_as(ctypes.POINTER(ctypes.c_float)), len(arr), float(k))
           for k in g['bs_keys']]
    assert got == list(g['bs_expected'])


def test_trajectory_4_3(golden):
    g = golden('trajectory')
    assert beam.earth_radius(7.0) == g['earth_radius_7']
    assert abs(g['earth_radius_7'] - 6368950.713464853) < 1e-6        # SURVEY 8(c) KAT
    conf = ocfg.make_config({'radar': {'coords': [46.0, 7.0, 500], 'frequency': 5.6,
                                       'range': 150000, 'radial_resolution': 300}})
    der = OK.Derived(conf)
    assert der.WAVELENGTH == g['wavelength']
    assert np.array_equal(der.RANGE_RADAR, g['range_vec'])
    for i, el in enumerate(g['elevations']):
        s, h, e = beam.trajectory_4_3(der.RANGE_RADAR, el, [46.0, 7.0, 500])
        for got, key in ((s, 's'), (h, 'h'), (e, 'e')):
            exp = g['%s_%d' % (key, i)]
            assert got.dtype == exp.dtype == np.float32
            assert np.array_equal(got, exp), (key, el)
    # SURVEY 8(c) known answers for elevation 1.0
    s, h, e = beam.trajectory_4_3(der.RANGE_RADAR, 1.0, [46.0, 7.0, 500])
    assert np.allclose([s[0], h[0], e[0]], [149.96828, 502.61917, 1.0010117], rtol=1e-6)
    assert np.allclose([s[499], h[499], e[499]], [149756.7, 4436.4688, 2.0104208], rtol=1e-6)


def test_quadrature_weights(golden):
    g = golden('quadrature')
    for ci in range(6):
        nh, nv, thr = g['case_%d' % ci]
        conf = ocfg.make_config({'radar': {'coords': [46.5, 7.5, 1000], '3dB_beamwidth': 1.3},
                                 'integration': {'nh_GH': int(nh), 'nv_GH': int(nv),
                                                 'weight_threshold': float(thr)}})
        ph, pv, w, keep = beam.gauss_hermite_subbeams(conf)
        pts, ws = [], []
        for i in range(len(ph)):
            for j in range(len(pv)):
                if keep[i, j]:
                    pts.append([ph[i] + 10.0, pv[j] + 5.0])
                    ws.append(w[i, j])
        assert np.array_equal(np.array(pts), g['pts_%d' % ci])
        assert np.array_equal(np.array(ws), g['w_%d' % ci])


def _lut_axis(h, freq, scheme):
    from cosmo_pol_amd import synthetic
    return synthetic.make_lut(h, freq, scheme, n_e=2, n_t=2).axes[2]


@pytest.mark.parametrize('h', ['R', 'S', 'G', 'I', 'mS', 'mG'])
def test_psd_one_moment(golden, h):
    g = golden('psd')
    QM, T_sol, fw, ksel = g['QM'], g['T_sol'], g['fw'], g['ksel']
    hy = psd.create_hydrometeor(h, '1mom')
    if h in ('R', 'S', 'G', 'I'):
        ax = _lut_axis(h, 5.6, '1mom')
        hy.d_min, hy.d_max = ax[0], ax[-1]
    if h in ('R', 'G'):
        hy.set_psd(QM)
    elif h in ('S', 'I'):
        hy.set_psd(T_sol, QM)
    elif h == 'mS':
        hy.set_psd(T_sol + 30, QM.astype(np.float64), fw)
    else:
        hy.set_psd(QM.astype(np.float64), fw)
    if h in ('mS', 'mG'):
        D = psd.vlinspace(hy.d_min, hy.d_max, 1024)
        N = hy.get_N(D)
        assert np.array_equal(hy.d_min, g[h + '_dmin']) and np.array_equal(hy.d_max, g[h + '_dmax'])
        np.testing.assert_allclose(hy.prop_factor, g[h + '_prop'], rtol=1e-13)
    else:
        N = hy.get_N(ax)
        assert str(N.dtype) == str(g[h + '_Ndtype'])
        np.testing.assert_allclose(np.asarray(hy.lambda_), g[h + '_lambda'], rtol=1e-14)
        np.testing.assert_allclose(np.asarray(hy.N0), g[h + '_N0'], rtol=1e-6)
    np.testing.assert_allclose(N[:, ksel], g[h + '_N'], rtol=1e-6)
    np.testing.assert_allclose(N.sum(axis=1), g[h + '_Nsum'], rtol=1e-6)
    v, n = hy.integrate_V()
    np.testing.assert_allclose(np.asarray(v), g[h + '_vint'], rtol=1e-6)
    np.testing.assert_allclose(np.asarray(n), g[h + '_nint'], rtol=1e-6)


@pytest.mark.parametrize('h', ['R', 'S', 'G', 'H', 'I'])
def test_psd_two_moment(golden, h):
    g = golden('psd')
    hy = psd.create_hydrometeor(h, '2mom')
    ax = _lut_axis(h, 13.6, '2mom')
    hy.d_min, hy.d_max = ax[0], ax[-1]
    hy.set_psd(g['2m_QN_' + h], g['QM'])
    N = hy.get_N(ax)
    np.testing.assert_allclose(np.asarray(hy.lambda_), g['2m_%s_lambda' % h], rtol=1e-13)
    np.testing.assert_allclose(np.asarray(hy.N0), g['2m_%s_N0' % h], rtol=1e-13)
    np.testing.assert_allclose(N[:, g['ksel']], g['2m_%s_N' % h], rtol=1e-6)
    np.testing.assert_allclose(N.sum(axis=1), g['2m_%s_Nsum' % h], rtol=1e-6)


def test_known_answer_constants():
    # SURVEY 8(c) known answers captured from the reference
    c = OK.C1
    assert abs(c.LAMBDA_FACTOR_R - 0.007631396756268) < 1e-15
    assert abs(c.N0_R - 1253.029102860344) < 1e-9
    assert abs(c.AM_R - 5.235987755982988e-07) < 1e-20
    assert abs(c.AV_R - 4.110960958218893) < 1e-12
    assert abs(c.LAMBDA_FACTOR_G - 0.002316328693023301) < 1e-16
    assert abs(c.AM_G - 8.500135482318533e-08) < 1e-20
    r = psd.Rain('1mom')
    r.set_psd(np.array([1e-4, 1e-3], dtype=np.float32))
    np.testing.assert_allclose(r.lambda_, [2.62033281, 1.57084822], rtol=2e-7)
    s = psd.Snow('1mom')
    s.set_psd(np.array([260, 270], dtype=np.float32), np.array([1e-4, 1e-3], dtype=np.float32))
    np.testing.assert_allclose(s.lambda_, [2.87137454, 0.93294953], rtol=2e-7)
    np.testing.assert_allclose(s.N0, [31149.85, 10684.664], rtol=2e-7)


def test_pol_from_sz(golden):
    g = golden('pol')
    conf = ocfg.make_config({'radar': {'coords': [46.0, 7.0, 500], 'frequency': 5.6}})
    res = scatter.pol_from_sz(g['sz'], conf)
    for name, r in zip(['ZH', 'ZV', 'ZDR', 'RHOHV', 'KDP', 'AH', 'AV', 'DELTA_HV'], res):
        assert r.dtype == g[name].dtype
        _cases.assert_close_nan(r, g[name], rtol=1e-6, name=name)
    # SURVEY 8(c) known answer
    sz = (np.arange(1, 13) * 1e-3 * np.array([1, -1, 1, 2, 3, -1, 1, 2, .5, .1, .7, .2]))[None]
    zh, zv, zdr, rho, kdp, ah, av, dhv = [float(x[0]) for x in scatter.pol_from_sz(sz, conf)]
    np.testing.assert_allclose([zh, zv, zdr, rho, kdp, ah, av, dhv],
                               [1450.668721082402, 1813.3359013530028, 0.8, 3.758324094593227,
                                0.00981533861328742, 0.001115998838652, 0.000464999516105,
                                -2.7445132083648107], rtol=1e-12)


def test_lut_bins(golden):
    g = golden('lut_lookup')
    from cosmo_pol_amd import synthetic
    for h, second in (('R', 't'), ('mS', 'wc')):
        L = _cases.as_oracle_lut(synthetic.make_lut(h, 5.6, '1mom', n_e=5))
        code = L.bin_index('e', g['e']) * 1000 + L.bin_index(second, g[second])
        assert np.array_equal(code, g[h + '_code'].astype(int))


@pytest.mark.parametrize('name', list(_cases.RADIAL_CASES))
def test_radial_end_to_end(golden, name):
    g = golden('radial_' + name)
    conf, az, el, cube, luts, _ = _cases.radial_case(name)
    subs = beam.interpolate_radial(cube, conf, az, el)
    assert len(subs) == int(g['n_sub'])
    # every fixture carries the sub-beam weights / nodes (regenerated in round 2)
    ng = g['quad_w'].shape[1]
    got_w = np.array([np.broadcast_to(sb.quad_weight, (ng,)) for sb in subs])
    np.testing.assert_allclose(got_w, g['quad_w'], rtol=1e-13, atol=1e-300)
    np.testing.assert_allclose(np.array([sb.quad_pt for sb in subs]), g['quad_pts'], rtol=0, atol=1e-12)
    c = subs[int(len(subs) / 2)]
    for tag, sb in (('c', c), ('f', subs[0])):
        for n in sb.values:
            exp = g['sub%s_%s' % (tag, n)]
            got = np.asarray(sb.values[n])
            assert got.dtype == exp.dtype, n
            assert np.array_equal(got, exp, equal_nan=True), (tag, n)
        assert np.array_equal(sb.mask, g['sub%s_mask' % tag])
        assert np.array_equal(sb.dist_profile, g['sub%s_s' % tag])
        assert np.array_equal(sb.heights_profile, g['sub%s_h' % tag])
        assert np.array_equal(sb.elev_profile, g['sub%s_e' % tag])
        np.testing.assert_allclose(sb.lats_profile, g['sub%s_lats' % tag], rtol=0, atol=1e-12)
        np.testing.assert_allclose(sb.lons_profile, g['sub%s_lons' % tag], rtol=0, atol=1e-12)
    integ = beam.integrate_subbeams(subs)
    for n in integ.values:
        _cases.assert_close_nan(integ.values[n], g['model_' + n], rtol=1e-12, name='model_' + n)
    assert np.array_equal(integ.mask, g['model_mask'])
    obs = scatter.radar_observables(subs, {h: _cases.as_oracle_lut(l) for h, l in luts.items()}, conf)
    for n in obs.values:
        # libm / SIMD differences between hosts: allow a few f32 ulps
        _cases.assert_close_nan(obs.values[n], g['obs_' + n], rtol=2e-6, atol=1e-30, name=n)
    assert np.array_equal(obs.mask, g['obs_mask'])
    # sensitivity cut, list-of-lists branch (what get_PPI / get_RHI use; spectrum bin by bin)
    n_cut = 0
    cut = scatter.cut_at_sensitivity([[obs]], conf)[0][0]
    for n in cut.values:
        _cases.assert_close_nan(cut.values[n], g['cutll_' + n], rtol=2e-6, atol=1e-30, name='cut:' + n)
        n_cut += int(np.isnan(g['cutll_' + n]).sum() - np.isnan(g['obs_' + n]).sum())
    if 'DSPECTRUM' in cut.values:
        assert n_cut > 0, 'the per-bin spectrum cut was not exercised'
        # ... and it is NOT the gate mask: bins survive at gates, bins vanish at kept gates
        kept = np.isfinite(g['cutll_ZH'])
        assert np.isnan(g['cutll_DSPECTRUM'][kept]).sum() > np.isnan(g['obs_DSPECTRUM'][kept]).sum()


def test_aliasing(golden):
    g = golden('aliasing')
    for i in range(3):
        got = scatter.aliasing(g['v'].copy(), float(g['nyq_%d' % i]))
        np.testing.assert_allclose(got, g['folded_%d' % i], rtol=0, atol=1e-12)
        assert np.all(np.abs(got) <= float(g['nyq_%d' % i]) + 1e-9)


def test_interp1d_restatement_equals_scipy():
    """oracle/cosmo_pol_oracle/spectrum.py::interp1d_linear restates what hydrometeors.py:494-500 asks of SciPy
    (interp1d, linear, bounds_error=False, fill_value=nan, assume_sorted=False): same bits on sorted, unsorted and
    out-of-range queries, repeated abscissae aside (the fall-speed table of a melting species is strictly increasing)."""
    from scipy.interpolate import interp1d
    from cosmo_pol_oracle import spectrum as SP
    rng = np.random.default_rng(20261005)
    for n in (2, 3, 17, 1024):
        x = np.cumsum(rng.uniform(1e-3, 1.0, n))
        y = rng.normal(size=n)
        perm = rng.permutation(n)
        q = np.concatenate([rng.uniform(x[0] - 1, x[-1] + 1, 500), x, [x[0], x[-1], np.nan]])
        for xs, ys in ((x, y), (x[perm], y[perm])):
            ref = interp1d(xs, ys, bounds_error=False, fill_value=np.nan, assume_sorted=False, copy=False)(q)
            got = SP.interp1d_linear(xs, ys, q)
            assert np.array_equal(ref, got, equal_nan=True), n
