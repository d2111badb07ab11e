"""GPU parity tests (run with -m gpu on an MI355X): the HIP path, called through
the C ABI (ctypes -> libcosmo_pol_hip.so), against
  * the golden vectors produced by the reference itself (tests/golden), and
  * the CPU oracle (oracle/) on the same seeded inputs.

Tolerances (BASELINE.json north_star): gate / level / LUT-bin indices, masks and
interpolated model variables are compared bit-exactly; polarimetric variables
to 1e-5 relative.  For the two differences of near-equal float32-stored sums
(KDP ~ sz10-sz8, and PHIDP/DELTA_HV built on it) the reference's own float32
store (quirk Q4) quantises each operand to 6e-8 relative, so the honest
comparison scale is the magnitude of the operands: |dKDP| <= 1e-5 * c *
(|sz8| + |sz10|) -- written out below.
"""
import numpy as np
import pytest

import _cases
from cosmo_pol_oracle import beam, scatter
from cosmo_pol_oracle import config as ocfg

pytestmark = pytest.mark.gpu

RTOL = 1e-5


def _operator(conf_over, luts, cube, output_variables='all'):
    from cosmo_pol_amd import RadarOperator
    op = RadarOperator(config=conf_over, luts=luts, output_variables=output_variables)
    op.load_model_arrays(cube['data'], cube['zlevels'], cube['proj_info'], cube['resolution'])
    return op


def test_native_library_loaded():
    from cosmo_pol_amd import _native
    lib = _native.load_library()
    assert lib is not None
    ctx = _native.Context(0)
    ctx.close()


def test_gate_kernel_bit_exact(golden):
    """cpol_interp_points == compiled reference C (get_all_radar_pts)."""
    from cosmo_pol_amd import _native
    g = golden('gate_kernel')
    ctx = _native.Context(0)
    zl = g['zlevels']
    nz, ny, nx = zl.shape
    llc, res = g['llc'], g['res']
    urc = np.array([llc[0] + res[0] * (nx - 1), llc[1] + res[1] * (ny - 1)], dtype=np.float32)
    rng = np.random.default_rng(1)
    second = rng.normal(size=zl.shape).astype(np.float32)
    ctx.stage_model([g['data'], second], zl, llc, urc, res, [-43.0, 10.0])
    out = ctx.interp_points(g['coords'], g['heights'])
    exp = g['expected']
    assert np.array_equal(np.isnan(out[0]), np.isnan(exp))
    ok = ~np.isnan(exp)
    assert np.array_equal(out[0][ok].view(np.uint32), exp[ok].view(np.uint32))
    # second variable: against the oracle's C twin
    exp2 = beam.get_all_radar_pts(g['coords'], g['heights'], second, zl, llc, res)
    assert np.array_equal(out[1], exp2, equal_nan=True)
    ctx.close()


def _record_parity(case, var, got, ref, sz_ref=None, conf=None):
    """Worst PURE relative deviation |got - ref| / |ref| of a variable, the number of gates
    above 1e-5 and (KDP, PHIDP, DELTA_HV: differences of float32-stored sums) the smallest
    cancellation factor (|sz8| + |sz10|) / |sz10 - sz8| among those gates.  Appended to
    gpurun_out/parity_records.jsonl (the table of DESIGN.md section 4) and printed."""
    import json
    import os
    got = np.asarray(got, dtype=np.float64).ravel()
    ref = np.asarray(ref, dtype=np.float64).ravel()
    ok = np.isfinite(ref) & np.isfinite(got) & (ref != 0)
    rel = np.zeros(ref.shape)
    rel[ok] = np.abs(got[ok] - ref[ok]) / np.abs(ref[ok])
    above = rel > RTOL
    rec = {'case': case, 'var': var, 'n': int(ok.sum()), 'worst_rel': float(rel.max()) if ok.any() else 0.0,
           'n_above_1e-5': int(above.sum())}
    if sz_ref is not None and above.any() and rel.shape[0] == sz_ref.shape[0]:
        with np.errstate(divide='ignore', invalid='ignore'):
            canc = (np.abs(sz_ref[:, 8]) + np.abs(sz_ref[:, 10])) / np.abs(sz_ref[:, 10] - sz_ref[:, 8])
        rec['min_cancellation_above'] = float(np.nanmin(canc[above]))
        rec['worst_rel_over_cancellation'] = float(np.nanmax(rel[above] / canc[above]))
    print('PARITY', json.dumps(rec))
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'gpurun_out')
    try:
        os.makedirs(out, exist_ok=True)
        with open(os.path.join(out, 'parity_records.jsonl'), 'a') as f:
            f.write(json.dumps(rec) + '\n')
    except OSError:
        pass
    return rec


def _pol_tolerances(name, obs_ref, sz_ref, conf):
    """atol per variable following the module docstring."""
    from cosmo_pol_oracle import constants as OK
    wl = OK.Derived(conf).WAVELENGTH
    res_km = conf['radar']['radial_resolution'] / 1000.
    kdp_scale = 1e-3 * (180.0 / np.pi) * wl * (np.abs(sz_ref[:, 8]) + np.abs(sz_ref[:, 10]))
    kdp_scale = np.nan_to_num(kdp_scale)
    if name == 'KDP':
        return RTOL * kdp_scale
    if name in ('PHIDP',):
        return RTOL * (np.cumsum(2 * kdp_scale) * res_km + np.pi)
    if name == 'DELTA_HV':
        return RTOL * np.pi
    return 0.0


@pytest.mark.parametrize('name', list(_cases.RADIAL_CASES))
def test_radial_vs_reference_golden_and_oracle(golden, name):
    g = golden('radial_' + name)
    conf, az, el, ocube, luts, cube = _cases.radial_case(name)
    over = _cases.gen_golden.radial_case_inputs(name)[0]
    op = _operator(over, luts, cube)
    op._ctx.enable_debug(True)
    res = op.simulate_rays([az], [el], apply_sensitivity=False)
    n_gates = res['ZH'].shape[1]

    # ---- oracle on the same inputs ----
    subs = beam.interpolate_radial(ocube, conf, az, el)
    n_sub = len(subs)
    assert res['n_sub'] == n_sub == int(g['n_sub'])
    oobs = scatter.radar_observables(subs, {h: _cases.as_oracle_lut(l) for h, l in luts.items()},
                                     conf, return_sz=True)

    # ---- bit-exact: ray paths, interpolated variables, masks, elevation ----
    n_v = len(np.unique([round(sb.quad_pt[1], 12) for sb in subs]))
    names = op._staged_vars
    n_sbg = n_sub * n_gates
    vals = op._ctx.debug_read('sub_values', (len(names), n_sub, n_gates), np.float32)
    mask = op._ctx.debug_read('sub_mask', (n_sub, n_gates), np.int8)
    elev = op._ctx.debug_read('sub_elev', (n_sub, n_gates), np.float32)
    coords = op._ctx.debug_read('sub_coords', (n_sub, n_gates, 2), np.float32)
    for s, sb in enumerate(subs):
        lats, lons, rc = beam.gate_coordinates(ocube, conf['radar']['coords'], sb.quad_pt[0],
                                               sb.dist_profile)
        assert np.array_equal(coords[s], rc), 'rotated coordinates differ (sub-beam %d)' % s
        assert np.array_equal(mask[s].astype(float), sb.mask)
        assert np.array_equal(elev[s], sb.elev_profile)
        for v, nm in enumerate(names):
            assert np.array_equal(vals[v, s], sb.values[nm], equal_nan=True), (nm, s)
    c = subs[int(n_sub / 2)]
    assert np.array_equal(res['dist'][0], c.dist_profile)
    assert np.array_equal(res['heights'][0], c.heights_profile)
    np.testing.assert_allclose(res['lats'][0], c.lats_profile, rtol=0, atol=1e-11)
    np.testing.assert_allclose(res['lons'][0], c.lons_profile, rtol=0, atol=1e-11)
    assert np.array_equal(res['mask'][0], oobs.mask)
    assert np.array_equal(res['mask'][0], g['obs_mask'])

    # ---- bit-exact: LUT bins (bucket keys) ----
    hl = ocfg.hydrometeor_list(conf)
    keys = op._ctx.debug_read('item_key', (len(hl), n_sub, n_gates), np.int32)
    base = 0
    n_valid = 0
    for j, h in enumerate(hl):
        L = _cases.as_oracle_lut(luts[h])
        n_t = L.value_table.shape[1]
        for s, sb in enumerate(subs):
            qm = np.asarray(sb.values['Q' + h + '_v'])
            with np.errstate(invalid='ignore'):
                valid = qm > 0
            if not np.isscalar(sb.quad_weight):      # scheme 'ml' (doppler_scatter.py:186-189)
                valid = np.logical_and(valid, sb.quad_weight > 0)
            assert np.array_equal(keys[j, s] >= 0, valid), (h, s)
            if valid.any():
                eb = L.bin_index('e', sb.elev_profile[valid])
                tb = (L.bin_index('wc', sb.values['fwet_' + h][valid]) if h in ('mS', 'mG')
                      else L.bin_index('t', sb.values['T'][valid]))
                assert np.array_equal(keys[j, s][valid], base + eb * n_t + tb), (h, s)
            n_valid += int(valid.sum())
        base += L.value_table.shape[0] * n_t
    cnt = op._ctx.counters()
    assert cnt.n_valid_items == n_valid
    if np.isscalar(subs[0].quad_weight):
        assert n_valid == int(g['n_valid'].sum())
    assert cnt.n_subbeam_gates == n_sbg

    # ---- 1e-5: integrated scattering entries and polarimetric variables ----
    szi = op._ctx.debug_read('sz_integ', (n_gates, len(hl), 12), np.float32)
    _cases.assert_close_nan(szi, oobs.sz_integ, rtol=RTOL, name='sz_integ')
    szt = op._ctx.debug_read('sz_total', (n_gates, 12), np.float32)
    _cases.assert_close_nan(szt, oobs.sz_total, rtol=RTOL, name='sz_total')
    for k in ['ZH', 'ZV', 'ZDR', 'RHOHV', 'KDP', 'ATT_H', 'ATT_V', 'DELTA_HV', 'PHIDP']:
        atol = _pol_tolerances(k, oobs, np.nan_to_num(oobs.sz_total.astype(np.float64)), conf)
        _cases.assert_close_nan(res[k][0], oobs.values[k], rtol=RTOL, atol=atol, name='oracle:' + k)
        _cases.assert_close_nan(res[k][0], g['obs_' + k], rtol=RTOL, atol=atol, name='golden:' + k)
        rec = _record_parity(name, k, res[k][0], g['obs_' + k],
                             np.nan_to_num(oobs.sz_total.astype(np.float64)), conf)
        # north_star: PURE 1e-5 relative on every polarimetric variable, K_DP included.  (Round 1
        # needed an operand-scaled tolerance for K_DP: 3 % of the gates were up to 3.4e-5 off,
        # every one of them through a 1-ulp difference of the float32 snow intercept -- NumPy's
        # float32 exp is not correctly rounded.  The device now looks that value up in a table
        # evaluated by the host's NumPy, cpol_stage_t_function.)
        assert rec['n_above_1e-5'] == 0, rec

    # ---- radial velocity (Doppler scheme 1): float64, terms of O(10 m/s) that may cancel ----
    assert 'RVEL' in res
    _cases.assert_close_nan(res['RVEL'][0], oobs.values['RVEL'], rtol=RTOL, atol=2e-4, name='oracle:RVEL')
    _cases.assert_close_nan(res['RVEL'][0], g['obs_RVEL'], rtol=RTOL, atol=2e-4, name='golden:RVEL')

    # ---- Doppler spectrum (scheme 3): float32 per-bin reflectivities, float64 sums ----
    if 'DSPECTRUM' in oobs.values:
        sp, osp = res['DSPECTRUM'][0], oobs.values['DSPECTRUM']
        assert sp.shape == osp.shape == g['obs_DSPECTRUM'].shape
        assert np.nansum(osp > 0) > 50, 'the spectrum was not exercised'
        atol = 1e-6 * np.nanmax(osp)
        _cases.assert_close_nan(sp, osp, rtol=2e-5, atol=atol, name='oracle:DSPECTRUM')
        _cases.assert_close_nan(sp, g['obs_DSPECTRUM'], rtol=2e-5, atol=atol, name='golden:DSPECTRUM')

    # ---- antenna-averaged model variables (integrate_radials) ----
    integ = beam.integrate_subbeams(subs)
    for i, nm in enumerate(names):
        _cases.assert_close_nan(res['model_vars'][i][0], integ.values[nm], rtol=1e-12,
                                name='model:' + nm)

    # ---- sensitivity cut on the device == the reference's list-of-lists branch (the Doppler
    # spectrum is censored bin by bin, doppler_scatter.py:839-850) ----
    op._ctx.enable_debug(False)
    cut = op.simulate_rays([az], [el], apply_sensitivity=True)
    for k in ['ZH', 'ZV', 'ZDR', 'RHOHV', 'KDP', 'ATT_H', 'ATT_V', 'DELTA_HV', 'PHIDP', 'RVEL', 'DSPECTRUM']:
        if ('cutll_' + k) not in g.files:
            continue
        gk = g['cutll_' + k]
        assert np.array_equal(np.isnan(cut[k][0]), np.isnan(gk)), 'cut pattern: ' + k
        assert np.array_equal(cut[k][0][~np.isnan(gk)], res[k][0][~np.isnan(gk)]), k
    # ---- the same radial without the debug reads: single-beam cases then take the fast path (k_gate1: PSD
    # parameters, table evaluation and get_pol_from_sz in one kernel), the others the general sequence without
    # its debug stores -- every output must carry the same bits as the run above ----
    plain = op.simulate_rays([az], [el], apply_sensitivity=False)
    for k, v in res.items():
        if isinstance(v, np.ndarray):
            assert np.array_equal(plain[k], v, equal_nan=True), 'fast path / no debug: ' + k
    op.close()


def test_small_ppi_vs_oracle_and_sensitivity():
    """A 24-ray PPI with 3x3 sub-beams, melting + ice, sensitivity cut on."""
    name = 'c4_subbeams'
    conf, _, _, ocube, luts, cube = _cases.radial_case(name)
    over = _cases.gen_golden.radial_case_inputs(name)[0]
    over = {k: dict(v) for k, v in over.items()}
    over['integration'] = {'nh_GH': 3, 'nv_GH': 3, 'weight_threshold': 1.}
    over['radar']['sensitivity'] = [35., 10000]
    conf = ocfg.make_config(over)
    op = _operator(over, luts, cube, output_variables='only_radar')
    azs = np.arange(0, 360, 15.)
    scan = op.get_PPI(elevations=[4.0], azimuths=azs)
    assert scan.nsweeps == 1 and scan.fields['ZH']['data'].shape == (24, len(op.constants.RANGE_RADAR))
    olut = {h: _cases.as_oracle_lut(l) for h, l in luts.items()}
    n_cut = 0
    for r, az in enumerate(azs):
        subs = beam.interpolate_radial(ocube, conf, az, 4.0)
        o = scatter.radar_observables(subs, olut, conf, return_sz=True)
        szt = np.nan_to_num(o.sz_total.astype(np.float64))
        before = np.isfinite(o.values['ZH']).sum()
        scatter.cut_at_sensitivity([o], conf)
        n_cut += before - np.isfinite(o.values['ZH']).sum()
        raw = scan.raw[0]['fields']
        for k in ['ZH', 'ZV', 'ZDR', 'RHOHV', 'KDP', 'ATT_H', 'ATT_V', 'DELTA_HV', 'PHIDP']:
            atol = _pol_tolerances(k, o, szt, conf)
            _cases.assert_close_nan(raw[k][r], o.values[k], rtol=RTOL, atol=atol,
                                    name='%s az=%g' % (k, az))
        _cases.assert_close_nan(raw['RVEL'][r], o.values['RVEL'], rtol=RTOL, atol=2e-4,
                                name='RVEL az=%g' % az)
    assert n_cut > 0, 'the sensitivity cut was not exercised'
    # dB convention of the packaged scan
    zh_db = scan.get_field(0, 'ZH')
    with np.errstate(invalid='ignore', divide='ignore'):
        exp = 10 * np.log10(scan.raw[0]['fields']['ZH'])
    assert np.allclose(zh_db.filled(np.nan), exp, equal_nan=True)
    op.close()


def test_domain_error_is_index_error():
    name = 'c2_rsg'
    _, _, _, _, luts, cube = _cases.radial_case(name)
    over = _cases.gen_golden.radial_case_inputs(name)[0]
    over = {k: dict(v) for k, v in over.items()}
    over['radar']['range'] = 150000          # leaves the 1.1 deg test cube
    op = _operator(over, luts, cube, output_variables='only_radar')
    with pytest.raises(IndexError):
        op.simulate_rays([10.0], [1.0])
    op.close()


def test_missing_model_and_bad_config():
    from cosmo_pol_amd import RadarOperator
    with pytest.raises(ValueError):
        RadarOperator(config={'radar': {'frequency': 5.6}}, luts={})      # coords mandatory
