"""GPU tests of the spaceborne (GPM-DPR, BASELINE config 5) and host-supplied
ray-path geometries.  The reference's GPM branch is dead as shipped (SURVEY.md
3.4): parity is against the oracle's restatement of the intended behaviour
(oracle/cosmo_pol_oracle/gpm.py) -- no reference pin exists for config 5."""
import numpy as np
import pytest

import _cases
from cosmo_pol_amd import gpm, synthetic
from cosmo_pol_oracle import beam, scatter
from cosmo_pol_oracle import config as ocfg
from cosmo_pol_oracle import gpm as ogpm

pytestmark = pytest.mark.gpu
RTOL = 1e-5
HYD_2MOM = ['R', 'S', 'G', 'H']


BAND = {'Ku': (13.6, 125), 'Ka': (35.6, 250)}      # constants/global_constants.py:152-159


@pytest.fixture(scope='module')
def luts_band():
    cache = {}

    def get(band):
        if band not in cache:
            cache[band] = {h: synthetic.make_lut(h, BAND[band][0], '2mom') for h in HYD_2MOM}
        return cache[band]
    return get


def _swath():
    return gpm.synthetic_swath(n_scans=3, n_rays=5, cross_track_deg=4.0, scan_spacing_m=6000.0)


@pytest.mark.parametrize('band', ['Ku', 'Ka'])
def test_gpm_swath_vs_oracle(luts_band, band):
    """Both halves of the dual-frequency swath of BASELINE config 5: KuPR (13.6 GHz, 125 m
    gates) and KaPR (35.6 GHz, 250 m gates; radar_operator.py:577-588)."""
    from cosmo_pol_amd import RadarOperator
    luts_ku = luts_band(band)
    freq, res_m = BAND[band]
    from test_gpu_parity import _pol_tolerances
    cube = synthetic.small_test_cube(hydrometeors=('R', 'S', 'G'), two_moment=True,
                                     **_cases.gen_golden.CUBE_KW)
    base = {'radar': {'coords': [46.5, 7.5, 1000], 'frequency': 5.6, 'K_squared': 0.93},
            'microphysics': {'scheme': '2mom', 'with_ice_crystals': 0, 'with_melting': 0},
            'integration': {'nh_GH': 1, 'nv_GH': 3}}
    lut_5_6 = {h: synthetic.make_lut(h, 5.6, '2mom', n_e=2, n_t=2) for h in HYD_2MOM}

    def provider(hl, freq, scheme):     # noqa: F811  (freq here: the frequency being loaded)
        assert scheme == '2mom'
        return {h: (luts_ku if freq == BAND[band][0] else lut_5_6)[h] for h in hl}
    op = RadarOperator(config=base, luts=provider, output_variables='only_radar')
    op.load_model_arrays(cube['data'], cube['zlevels'], cube['proj_info'], cube['resolution'])
    sw = _swath()
    op._ctx.enable_debug(True)
    out = op.get_GPM_swath(sw, band)
    assert op.config['radar']['frequency'] == 5.6          # configuration restored
    N, M = sw['Latitude'].shape
    assert out.data['ZH'].shape[:2] == (N, M) and out.band == band

    # ---- oracle: same rays, per-ray satellite site ----
    over = {k: dict(v) for k, v in base.items()}
    over['radar'].update(frequency=freq, radial_resolution=res_m, sensitivity=12.0, type='GPM')
    over['radar']['3dB_beamwidth'] = 0.5
    conf = ocfg.make_config(over)
    order = _cases.ORDER_2MOM
    ocube = beam.ModelCube({n: cube['data'][n].copy() for n in order}, cube['zlevels'],
                           cube['proj_info'], cube['resolution'], order)
    olut = {h: _cases.as_oracle_lut(luts_ku[h]) for h in HYD_2MOM}
    az, el, rng, sat = ogpm.swath_angles(sw)
    np.testing.assert_allclose(out.azimuths, az, rtol=0, atol=1e-9)
    np.testing.assert_allclose(out.elevations, el, rtol=0, atol=1e-12)
    np.testing.assert_allclose(out.ranges, rng, rtol=1e-15)
    raw = out.raw
    n_valid_total = 0
    for idx in range(N * M):
        i, j = divmod(idx, M)
        subs, k0, n = ogpm.interpolate_swath_ray(ocube, conf, out.azimuths[i, j],
                                                 out.elevations[i, j], out.ranges[i, j], sat[i])
        assert n == out.n_kept[i, j], (idx, n, out.n_kept[i, j])
        c = subs[int(len(subs) / 2)]
        assert np.array_equal(raw['dist'][idx, :n], c.dist_profile)
        assert np.array_equal(raw['heights'][idx, :n], c.heights_profile)
        assert np.all(np.isnan(raw['dist'][idx, n:]))
        o = scatter.radar_observables(subs, olut, conf, return_sz=True, doppler=False)
        szt = np.nan_to_num(o.sz_total.astype(np.float64))
        scatter.cut_at_sensitivity([o], conf)
        assert np.array_equal(raw['mask'][idx, :n], o.mask)
        for k in ['ZH', 'ZV', 'ZDR', 'RHOHV', 'KDP', 'ATT_H', 'ATT_V', 'DELTA_HV', 'PHIDP']:
            atol = _pol_tolerances(k, o, szt, conf)
            _cases.assert_close_nan(raw[k][idx, :n], o.values[k], rtol=RTOL, atol=atol,
                                    name='%s ray %d' % (k, idx))
            assert np.all(np.isnan(raw[k][idx, n:]))
        n_valid_total += int(np.isfinite(o.values['ZH']).sum())
    assert n_valid_total > (200 if band == 'Ku' else 100)
    # packaging: beams start at the ground, gates under the topography removed
    assert np.isfinite(out.lats[0, 0, 0]) and out.bin_surface.shape == (N, M)
    op.close()


def test_host_supplied_paths_equal_builtin_model():
    """CPOL_GEOM_HOST_PATHS with the 4/3-earth paths computed on the host gives
    bit-identical results to the built-in model (the plumbing the Zeng & Blahak
    ODE refraction uses)."""
    name = 'c4_subbeams'
    conf, az, el, ocube, luts, cube = _cases.radial_case(name)
    over = _cases.gen_golden.radial_case_inputs(name)[0]
    from cosmo_pol_amd import RadarOperator
    op = RadarOperator(config=over, luts=luts, output_variables='only_radar')
    op.load_model_arrays(cube['data'], cube['zlevels'], cube['proj_info'], cube['resolution'])
    azs, els = np.array([az, az + 40.0]), np.array([el, el + 1.5])
    ref = op.simulate_rays(azs, els)
    from cosmo_pol_oracle import constants as OK
    pts_hor, pts_ver, w, keep = beam.gauss_hermite_subbeams(conf)
    rr = OK.Derived(conf).RANGE_RADAR
    paths = np.zeros((2, len(pts_ver), 3, len(rr)), dtype=np.float32)
    for r in range(2):
        for j, pt in enumerate(pts_ver):
            paths[r, j] = np.stack(beam.trajectory_4_3(rr, pt + els[r], conf['radar']['coords']))
    got = op.simulate_rays(azs, els, paths=paths)
    for k in ['ZH', 'ZDR', 'KDP', 'PHIDP', 'RHOHV', 'dist', 'heights', 'mask']:
        assert np.array_equal(got[k], ref[k], equal_nan=True), k
    op.close()


def test_refraction_scheme_2_vs_oracle():
    """refraction/scheme: 2 -> host ODE ray paths -> GPU; against the oracle fed
    with its own restatement of the ODE (parity unpinned vs the reference, whose
    _ref_ODE raises under NumPy >= 1.24)."""
    from cosmo_pol_amd import RadarOperator
    from cosmo_pol_oracle import refraction_ode
    from cosmo_pol_oracle import constants as OK
    from test_gpu_parity import _pol_tolerances
    name = 'c2_rsg'
    conf, az, el, ocube, luts, cube = _cases.radial_case(name)
    over = _cases.gen_golden.radial_case_inputs(name)[0]
    over = {k: dict(v) for k, v in over.items()}
    over['refraction'] = {'scheme': 2}
    over['integration'] = {'nh_GH': 1, 'nv_GH': 3}
    conf = ocfg.make_config(over)
    Nf = _cases.gen_golden.refractivity_field(cube)
    data = dict(cube['data'])
    data['N'] = Nf
    op = RadarOperator(config=over, luts=luts, output_variables='only_radar')
    op.load_model_arrays(data, cube['zlevels'], cube['proj_info'], cube['resolution'])
    res = op.simulate_rays([az], [el], apply_sensitivity=False)
    rr = OK.Derived(conf).RANGE_RADAR
    pts_hor, pts_ver, w, keep = beam.gauss_hermite_subbeams(conf)
    trajs = [refraction_ode.trajectory_ode(rr, pt + el, conf['radar']['coords'], Nf, cube['zlevels'],
                                           cube['proj_info'], cube['resolution']) for pt in pts_ver]
    subs = beam.interpolate_radial(ocube, conf, az, el, trajs=trajs)
    o = scatter.radar_observables(subs, {h: _cases.as_oracle_lut(l) for h, l in luts.items()}, conf,
                                  return_sz=True)
    c = subs[int(len(subs) / 2)]
    # the ray paths of product and oracle are the same bits (tests/test_refraction_cpu.py), so everything
    # downstream holds the contract's 1e-5 (parity with the REFERENCE stays unpinned: its _ref_ODE raises
    # inside scipy.odeint under NumPy >= 1.24, oracle/gen_golden.py)
    assert np.array_equal(res['heights'][0], c.heights_profile) and np.array_equal(res['dist'][0], c.dist_profile)
    assert np.array_equal(res['mask'][0], o.mask)
    szt = np.nan_to_num(o.sz_total.astype(np.float64))
    for k in ['ZH', 'ZV', 'ZDR', 'RHOHV', 'KDP', 'ATT_H', 'ATT_V', 'DELTA_HV', 'PHIDP']:
        atol = _pol_tolerances(k, o, szt, conf)
        _cases.assert_close_nan(res[k][0], o.values[k], rtol=1e-5, atol=atol, name=k)
    # and the paths really differ from the 4/3 model
    over43 = {k: dict(v) for k, v in over.items()}
    over43['refraction'] = {'scheme': 1}
    op.config = over43
    res43 = op.simulate_rays([az], [el], apply_sensitivity=False)
    assert not np.array_equal(res43['heights'], res['heights'])
    op.close()


def test_rvel_aliasing_vs_oracle(tmp_path):
    """radar/nyquist_velocity file -> per-ray Nyquist velocity -> folded RVEL."""
    from cosmo_pol_amd import RadarOperator
    name = 'c2_rsg'
    conf, az, el, ocube, luts, cube = _cases.radial_case(name)
    over = _cases.gen_golden.radial_case_inputs(name)[0]
    over = {k: dict(v) for k, v in over.items()}
    fn = tmp_path / 'nyq.txt'
    fn.write_text('elevation,azimuth,nyquist\n0.5,0,0.3\n4.0,0,3.3\n4.0,180,1.1\n9.0,0,6.\n')
    over['radar']['nyquist_velocity'] = str(fn)
    op = RadarOperator(config=over, luts=luts, output_variables='only_radar')
    op.load_model_arrays(cube['data'], cube['zlevels'], cube['proj_info'], cube['resolution'])
    azs, els = np.array([az, 20.0]), np.array([el, 0.7])
    res = op.simulate_rays(azs, els, apply_sensitivity=False)
    olut = {h: _cases.as_oracle_lut(l) for h, l in luts.items()}
    folded = 0
    for r, nyq in enumerate([1.1, 0.3]):
        subs = beam.interpolate_radial(ocube, conf, azs[r], els[r])
        plain = scatter.radar_observables(subs, olut, conf).values['RVEL']
        subs = beam.interpolate_radial(ocube, conf, azs[r], els[r])
        o = scatter.radar_observables(subs, olut, conf, nyquist=nyq)
        _cases.assert_close_nan(res['RVEL'][r], o.values['RVEL'], rtol=1e-5, atol=2e-4, name='RVEL')
        assert np.nanmax(np.abs(res['RVEL'][r])) <= nyq + 1e-9
        folded += int(np.nansum(np.abs(plain) > nyq))
    assert folded > 10, 'aliasing was not exercised'
    op.close()


def test_doppler_spectrum_one_moment_ice_subbeams_vs_oracle(tmp_path):
    """Doppler scheme 3 beyond the two reference goldens: 1-moment ice (its normalised N0
    comes from the PSD kernel), three sub-beams, attenuation on, aliasing on."""
    from cosmo_pol_amd import RadarOperator
    name = 'c3_melt_ice'
    over = _cases.gen_golden.radial_case_inputs(name)[0]
    over = {k: dict(v) for k, v in over.items()}
    over['microphysics'].update(with_melting=0, with_ice_crystals=1)
    over['doppler'] = {'scheme': 3}
    over['integration'] = {'nh_GH': 3, 'nv_GH': 1, 'weight_threshold': 1.}
    over['radar'].update(FFT_length=32, range=30000, radial_resolution=500)
    fn = tmp_path / 'nyq.txt'
    fn.write_text('elevation,azimuth,nyquist\n5.0,0,1.5\n')
    over['radar']['nyquist_velocity'] = str(fn)
    _, _, _, ocube, _, cube = _cases.radial_case(name)
    conf = ocfg.make_config({k: {kk: vv for kk, vv in v.items() if kk != 'nyquist_velocity'}
                             for k, v in over.items()})
    hl = ocfg.hydrometeor_list(conf)
    luts = {h: _cases.synthetic_lut(h, 5.6, '1mom') for h in hl}
    op = RadarOperator(config=over, luts=luts, output_variables='only_radar')
    op.load_model_arrays(cube['data'], cube['zlevels'], cube['proj_info'], cube['resolution'])
    azs, els = np.array([120.0, 300.0]), np.array([5.0, 9.0])
    res = op.simulate_rays(azs, els, apply_sensitivity=False)
    olut = {h: _cases.as_oracle_lut(l) for h, l in luts.items()}
    n_ice = 0
    for r in range(2):
        subs = beam.interpolate_radial(ocube, conf, azs[r], els[r])
        n_ice += int(sum(np.nansum(np.asarray(s.values['QI_v']) > 0) for s in subs))
        o = scatter.radar_observables(subs, olut, conf, nyquist=1.5)
        osp = o.values['DSPECTRUM']
        atol = 1e-6 * np.nanmax(osp)
        _cases.assert_close_nan(res['DSPECTRUM'][r], osp, rtol=2e-5, atol=atol, name='DSPECTRUM')
        _cases.assert_close_nan(res['RVEL'][r], o.values['RVEL'], rtol=1e-5, atol=2e-4, name='RVEL')
        assert np.nanmax(np.abs(res['RVEL'][r])) <= 1.5 + 1e-9
    assert n_ice > 20, 'no ice crystals in the test rays'
    op.close()


def test_c5_bench_size_swaths_sampled_rays_vs_oracle():
    """BASELINE configs[4] at the bench's size: Ku 200 x 49 and Ka 200 x 25 rays over the 2-moment
    bench cube (R,S,G,H,I, full-size tables) through get_GPM_swath -- what `bench.py --workload c5`
    times.  Eight sampled rays per band against the oracle (gate bookkeeping bit-exact, observables
    at 1e-5); no reference pin exists for config 5."""
    import bench
    from cosmo_pol_amd import RadarOperator
    from test_gpu_parity import _pol_tolerances
    hyd = ['R', 'S', 'G', 'H', 'I']
    conf0 = bench.bench_config(False, 'c5')
    cube = synthetic.make_cube(hydrometeors=tuple(hyd), two_moment=True, **synthetic.BENCH_GRID)
    sets = {}

    def provider(hl, freq, scheme):
        if freq not in sets:
            sets[freq] = synthetic.make_all_luts(hl, freq, scheme)
        return {h: sets[freq][h] for h in hl}
    op = RadarOperator(config=conf0, luts=provider, output_variables='only_radar')
    op.load_model_arrays(cube['data'], cube['zlevels'], cube['proj_info'], cube['resolution'])
    order = _cases.ORDER_2MOM
    ocube = beam.ModelCube({n: cube['data'][n] for n in order}, cube['zlevels'],
                           cube['proj_info'], cube['resolution'], order)
    for band, n_rays, cross in (('Ku', 49, 17.0), ('Ka', 25, 8.5)):
        freq, res_m = BAND[band]
        sw = gpm.synthetic_swath(n_scans=200, n_rays=n_rays, centre=(46.5, 7.5), cross_track_deg=cross,
                                 scan_spacing_m=3000.0)
        out = op.get_GPM_swath(sw, band)
        N, M = sw['Latitude'].shape
        assert (N, M) == (200, n_rays) and out.raw['ZH'].shape[0] == N * M
        over = {k: dict(v) for k, v in conf0.items()}
        over['radar'].update(frequency=freq, radial_resolution=res_m, sensitivity=12.0, type='GPM')
        over['radar']['3dB_beamwidth'] = 0.5
        conf = ocfg.make_config(over)
        olut = {h: _cases.as_oracle_lut(sets[freq][h]) for h in hyd}
        az, el, rng, sat = ogpm.swath_angles(sw)
        raw = out.raw
        # rays with the most precipitation along the swath centre and edges, spread over the scans
        finite = np.isfinite(raw['ZH']).sum(axis=1).reshape(N, M)
        picks = []
        for i in (3, 40, 77, 101, 133, 160, 181, 197):
            j = int(np.argmax(finite[i]))
            picks.append((i, j))
        n_valid_total = 0
        for i, j in picks:
            idx = i * M + j
            subs, k0, n = ogpm.interpolate_swath_ray(ocube, conf, az[i, j], el[i, j], rng[i, j], sat[i])
            assert n == out.n_kept[i, j]
            cc = subs[int(len(subs) / 2)]
            assert np.array_equal(raw['dist'][idx, :n], cc.dist_profile)
            assert np.array_equal(raw['heights'][idx, :n], cc.heights_profile)
            o = scatter.radar_observables(subs, olut, conf, return_sz=True, doppler=False)
            szt = np.nan_to_num(o.sz_total.astype(np.float64))
            scatter.cut_at_sensitivity([o], conf)
            assert np.array_equal(raw['mask'][idx, :n], o.mask)
            for k in ['ZH', 'ZV', 'ZDR', 'RHOHV', 'KDP', 'ATT_H', 'ATT_V', 'DELTA_HV', 'PHIDP']:
                atol = _pol_tolerances(k, o, szt, conf)
                _cases.assert_close_nan(raw[k][idx, :n], o.values[k], rtol=RTOL, atol=atol,
                                        name='%s %s ray (%d, %d)' % (band, k, i, j))
            n_valid_total += int(np.isfinite(o.values['ZH']).sum())
        assert n_valid_total > 200, (band, n_valid_total)
        # the lazily packed container: beams flipped to start at the ground
        i, j = picks[0]
        n = int(out.n_kept[i, j])
        zh = out.data['ZH'][i, j]
        keep = raw['mask'][i * M + j, :n] > -1
        assert np.array_equal(zh[:keep.sum()], raw['ZH'][i * M + j, :n][keep][::-1].astype(np.float64), equal_nan=True)
        assert np.all(zh[keep.sum():] == 0)
    op.close()


def test_swath_geometry_cache_follows_the_swath(luts_band):
    """get_GPM_swath keeps the swath-derived geometry (inverse geodesic, ray tables, first-gate pass) of
    the last few swaths, keyed by the CONTENT of the swath and the band: the same swath again gives the
    same bits without recomputing, another swath (or the other band) does not see a stale entry."""
    from cosmo_pol_amd import RadarOperator
    cube = synthetic.small_test_cube(hydrometeors=('R', 'S', 'G'), two_moment=True,
                                     **_cases.gen_golden.CUBE_KW)
    base = {'radar': {'coords': [46.5, 7.5, 1000], 'frequency': 5.6, 'K_squared': 0.93},
            'microphysics': {'scheme': '2mom', 'with_ice_crystals': 0, 'with_melting': 0},
            'integration': {'nh_GH': 1, 'nv_GH': 1}}
    lut_5_6 = {h: synthetic.make_lut(h, 5.6, '2mom', n_e=2, n_t=2) for h in HYD_2MOM}

    def provider(hl, freq, scheme):
        src = {13.6: luts_band('Ku'), 35.6: luts_band('Ka')}.get(freq, lut_5_6)
        return {h: src[h] for h in hl}
    op = RadarOperator(config=base, luts=provider, output_variables='only_radar')
    op.load_model_arrays(cube['data'], cube['zlevels'], cube['proj_info'], cube['resolution'])
    sw1 = _swath()
    sw2 = gpm.synthetic_swath(n_scans=3, n_rays=5, cross_track_deg=3.0, scan_spacing_m=5000.0, heading_deg=75.0)
    a1 = op.get_GPM_swath(sw1, 'Ku')
    assert len(op._gpm_cache) == 1
    a2 = op.get_GPM_swath(dict(sw1), 'Ku')                       # an equal swath in another dict: cache hit
    assert len(op._gpm_cache) == 1
    b = op.get_GPM_swath(sw2, 'Ku')
    c = op.get_GPM_swath(sw1, 'Ka')
    assert len(op._gpm_cache) == 3
    for k in ['ZH', 'KDP', 'lats', 'heights', 'mask']:
        assert np.array_equal(a1.raw[k], a2.raw[k], equal_nan=True), k
    assert not np.array_equal(a1.azimuths, b.azimuths)
    assert a1.raw['ZH'].shape != c.raw['ZH'].shape or not np.array_equal(a1.raw['ZH'], c.raw['ZH'], equal_nan=True)
    # a fresh operator (no cache) agrees with the cached answers
    op2 = RadarOperator(config=base, luts=provider, output_variables='only_radar')
    op2.load_model_arrays(cube['data'], cube['zlevels'], cube['proj_info'], cube['resolution'])
    b2 = op2.get_GPM_swath(sw2, 'Ku')
    for k in ['ZH', 'KDP', 'lats', 'heights', 'mask']:
        assert np.array_equal(b.raw[k], b2.raw[k], equal_nan=True), k
    # re-staging the model invalidates the entries (the first-gate pass depends on nothing of the cube, but
    # the key carries the staging serial so that no entry outlives its model)
    op.load_model_arrays(cube['data'], cube['zlevels'], cube['proj_info'], cube['resolution'])
    op.get_GPM_swath(sw1, 'Ku')
    assert len(op._gpm_cache) <= 4
    op.close()
    op2.close()


def test_single_beam_kernel_by_species_equals_by_gate_on_a_two_moment_swath(luts_band, monkeypatch):
    """k_gate1_species (one wavefront per species: the default of small sweeps) against k_gate1 (one thread per gate
    through all species: what a bench-size swath gets) on a 2-moment Ku swath with four species: every field bit for bit."""
    from cosmo_pol_amd import RadarOperator
    cube = synthetic.small_test_cube(hydrometeors=('R', 'S', 'G'), two_moment=True, **_cases.gen_golden.CUBE_KW)
    base = {'radar': {'coords': [46.5, 7.5, 1000], 'frequency': 5.6, 'K_squared': 0.93},
            'microphysics': {'scheme': '2mom', 'with_ice_crystals': 0, 'with_melting': 0},
            'integration': {'nh_GH': 1, 'nv_GH': 1}}
    lut_5_6 = {h: synthetic.make_lut(h, 5.6, '2mom', n_e=2, n_t=2) for h in HYD_2MOM}

    def provider(hl, freq, scheme):
        src = {13.6: luts_band('Ku'), 35.6: luts_band('Ka')}.get(freq, lut_5_6)
        return {h: src[h] for h in hl}
    sw = gpm.synthetic_swath(n_scans=6, n_rays=9, cross_track_deg=6.0, scan_spacing_m=5000.0)
    out = {}
    for mode in ('0', '2'):
        monkeypatch.setenv('CPOL_GATE1_SPECIES', mode)           # (read when the context is created)
        op = RadarOperator(config=base, luts=provider, output_variables='only_radar')
        op.load_model_arrays(cube['data'], cube['zlevels'], cube['proj_info'], cube['resolution'])
        out[mode] = op.get_GPM_swath(sw, 'Ku').raw
        c = op._ctx.counters()
        assert c.n_table_items > 500
        op.close()
    for k, v in out['0'].items():
        if isinstance(v, np.ndarray):
            assert np.array_equal(out['2'][k], v, equal_nan=True), k
    assert np.isfinite(out['0']['ZH']).sum() > 200


def test_doppler_spectrum_with_melting_at_the_largest_fft_length_vs_oracle():
    """FFT_length = 2048 -- the upper end of the reference's valid range (cfg.py:91) -- with the full 1-moment species set
    (R, S, G, mS, mG, I): 2 049 velocity bins x 6 species need 100 KB of LDS in k_spec_gate, beyond the default 64 KB per
    workgroup (gfx950: 160 KB per CU, asked for with hipFuncSetAttribute).  One ray through the melting layer against the
    oracle at the spectrum's tolerance."""
    import copy
    from cosmo_pol_amd import RadarOperator
    name = 'd3_melt_ice_sub'
    over = copy.deepcopy(_cases.gen_golden.radial_case_inputs(name)[0])
    over['radar'].update(FFT_length=2048, range=6000, radial_resolution=200)
    over['integration'].update(nh_GH=1, nv_GH=1)
    _, az, el, ocube, luts, cube = _cases.radial_case(name)
    conf = ocfg.make_config(over)
    op = RadarOperator(config=over, luts=luts, output_variables='only_radar', lanes=1)
    op.load_model_arrays(cube['data'], cube['zlevels'], cube['proj_info'], cube['resolution'])
    res = op.simulate_rays([az], [el], apply_sensitivity=False)
    subs = beam.interpolate_radial(ocube, conf, az, el)
    o = scatter.radar_observables(subs, {h: _cases.as_oracle_lut(l) for h, l in luts.items()}, conf)
    sp, osp = res['DSPECTRUM'][0], o.values['DSPECTRUM']
    assert sp.shape == osp.shape and sp.shape[1] == 2049
    assert np.nansum(osp > 0) > 500 and np.sum(np.asarray(subs[0].values['QmS_v']) > 0) > 3
    # With 2 049 velocity bins a bin spans one or two of the 1 024 table bins: the truncations (int)((D - D_min) / step) of
    # doppler_c.c sit on the last bit of the inverted diameters (NumPy's float32 sin of the elevation is
    # within an ulp, not correctly rounded), and ONE table bin may land in the neighbouring velocity bin.  The criterion of
    # tools/fuzz_parity.py: a handful of such bins at most, the power of every gate unchanged.
    atol = 1e-6 * np.nanmax(osp)
    bad = np.abs(sp - osp) > atol + 2e-5 * np.abs(osp)
    assert bad.sum() <= 4, 'DSPECTRUM: %d of %d bins differ' % (bad.sum(), bad.size)
    _cases.assert_close_nan(np.where(bad, 0.0, sp), np.where(bad, 0.0, osp), rtol=2e-5, atol=atol, name='DSPECTRUM')
    clean = ~bad.any(axis=1)
    _cases.assert_close_nan(np.nansum(sp, axis=1)[clean], np.nansum(osp, axis=1)[clean], rtol=2e-5, atol=atol, name='DSPECTRUM power')
    _cases.assert_close_nan(np.nansum(sp, axis=1), np.nansum(osp, axis=1), rtol=1e-4, atol=atol, name='DSPECTRUM power (all gates)')
    _cases.assert_close_nan(res['RVEL'][0][clean], o.values['RVEL'][clean], rtol=1e-5, atol=2e-4, name='RVEL')
    _cases.assert_close_nan(res['RVEL'][0], o.values['RVEL'], rtol=1e-3, atol=1e-2, name='RVEL (gates with a moved bin)')
    op.close()
