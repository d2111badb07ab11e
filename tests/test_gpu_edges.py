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


def test_items_outside_the_integral_tables_three_ways(monkeypatch):
    """Mass densities so small that the PSD slope leaves the tabulated range (lambda beyond the last accepted
    panel): such items go to the integrating kernels.  Three launch sequences must give the same bits:
    the general one with the counting sort (CPOL_RARE_DIRECT=0, CPOL_GATE1=0), the general one with the items
    listed directly as one-item work units (CPOL_GATE1=0) and the single-beam fused kernel, which defers the
    gates that hold such an item to k_final -- with one and with nine sub-beams, hundreds of items each.  Also the
    two-kernel form of the direct listing (CPOL_FUSE_CLASSIFY=0), the opt-in k_interp_gate1 (CPOL_FUSE_GATE1=1) and both forms of the
    single-beam kernel: one wavefront per species (k_gate1_species, the default of small sweeps) / one thread per gate (k_gate1),
    and the opt-in k_gate1_ray, which integrates such items in place (lanes 0..7 of the wavefront stand in for the eight
    wavefronts of the integrating kernel) -- with the range scans by k_scan_rays or inside the gate kernel."""
    import bench
    from cosmo_pol_amd import RadarOperator, synthetic
    hyds = ('R', 'S', 'G')
    cube = synthetic.small_test_cube(hydrometeors=hyds)
    data = {k: v.copy() for k, v in cube['data'].items()}
    # a wedge of the domain with rain / snow of 1e-15 .. 1e-19 kg m-3 (lambda far beyond the tables), the rest as it is
    ny, nx = data['QR_v'].shape[1:]
    yy, xx = np.meshgrid(np.arange(ny), np.arange(nx), indexing='ij')
    wedge = (xx > nx // 2) & (yy > ny // 2)
    for k, tiny in (('QR_v', 1e-16), ('QS_v', 3e-18)):
        f = data[k]
        f[:, wedge] = np.where(f[:, wedge] > 0, np.float32(tiny) * (1 + (np.arange(f.shape[0]) % 5))[:, None], 0).astype(np.float32)
    luts = synthetic.make_all_luts(hyds, 5.6, '1mom', n_e=8)
    az = np.arange(20.0, 70.0, 2.5)
    results = {}
    for n_gh in (1, 3):
        conf = bench.bench_config(True)
        conf['integration'].update(nh_GH=n_gh, nv_GH=n_gh)
        for mode, env in (('sorted', {'CPOL_RARE_DIRECT': '0', 'CPOL_GATE1': '0'}), ('direct', {'CPOL_GATE1': '0'}), ('default', {}),
                          ('direct_launch_per_flavour', {'CPOL_GATE1': '0', 'CPOL_PSD_RARE': '0'}),   # (round 4: k_psd_uniform alone instead of k_psd_rare)
                          ('direct_two_kernels', {'CPOL_GATE1': '0', 'CPOL_FUSE_CLASSIFY': '0'}),     # k_interp_sweep + k_classify
                          ('interp_gate1', {'CPOL_FUSE_GATE1': '1'}),                                  # k_interp_gate1 (opt-in)
                          ('gate1_one_thread', {'CPOL_GATE1_SPECIES': '0'}),       # k_gate1 instead of k_gate1_species (what large swaths get)
                          ('gate1_ray', {'CPOL_GATE1_RAY': '1'}),          # k_gate1_ray (off-table items integrated in place) + k_scan_rays: no
                                                                           # integrating launch, no k_final (opt-in: no faster, profiles/r5_variants.txt)
                          ('gate1_ray_ticket', {'CPOL_GATE1_RAY': '3'})):  # ... with the ray's scans inside the gate kernel (a ticket per ray)
            for k in ('CPOL_RARE_DIRECT', 'CPOL_GATE1', 'CPOL_FUSE_CLASSIFY', 'CPOL_FUSE_GATE1', 'CPOL_GATE1_SPECIES', 'CPOL_GATE1_RAY', 'CPOL_PSD_RARE'):
                monkeypatch.delenv(k, raising=False)
            for k, v in env.items():
                monkeypatch.setenv(k, v)                    # (read when the context is created)
            op = RadarOperator(config=conf, luts=luts, output_variables='only_radar')
            op.load_model_arrays(data, cube['zlevels'], cube['proj_info'], cube['resolution'])
            res = op.simulate_rays(az, np.full(len(az), 2.0))
            c = op._ctx.counters()
            n_off = int(c.n_valid_items - c.n_table_items)
            assert n_off > 200, (mode, n_off)
            assert int(c.n_work_units) == (n_off if mode != 'sorted' else int(c.n_work_units)) and c.n_work_units > 0
            results[(n_gh, mode)] = (res, n_off)
            op.close()
        ref, n_ref = results[(n_gh, 'sorted')]
        for mode in ('direct', 'default', 'direct_launch_per_flavour', 'direct_two_kernels', 'interp_gate1', 'gate1_one_thread', 'gate1_ray', 'gate1_ray_ticket'):
            got, n_got = results[(n_gh, mode)]
            assert n_got == n_ref
            for k in ('ZH', 'ZV', 'ZDR', 'KDP', 'RHOHV', 'PHIDP', 'DELTA_HV', 'ATT_H', 'ATT_V', 'RVEL', 'mask'):
                assert np.array_equal(got[k], ref[k], equal_nan=True), (n_gh, mode, k)
        assert np.isfinite(ref['ZH']).sum() > 500


def test_fused_kernel_with_melting_species_equals_the_general_sequence(monkeypatch):
    """CPOL_GATE1=2 takes the single-beam fused kernel also when melting species are present (k_gate1<true>: the
    2-D walk inside, results to their lanes through LDS; not the default: slower there).  Same bits as the
    general sequence, RVEL with the ice total credited to one gate per ray included."""
    import bench
    from cosmo_pol_amd import RadarOperator, synthetic
    hyds = ['R', 'S', 'G', 'mS', 'mG', 'I']
    conf = bench.bench_config(True, 'c3')
    cube = synthetic.small_test_cube(hydrometeors=('R', 'S', 'G', 'I'))
    luts = synthetic.make_all_luts(hyds, 5.6, '1mom', n_e=8)
    az = np.arange(0.0, 360.0, 7.5)
    out = {}
    for mode in ('0', '2', '2+interp'):
        monkeypatch.setenv('CPOL_GATE1', mode[0])               # (read when the context is created)
        monkeypatch.setenv('CPOL_FUSE_GATE1', '1' if mode == '2+interp' else '0')      # k_interp_gate1<true>
        op = RadarOperator(config=conf, luts=luts, output_variables='all')
        op.load_model_arrays(cube['data'], cube['zlevels'], cube['proj_info'], cube['resolution'])
        out[mode] = op.simulate_rays(az, np.full(len(az), 3.0))
        c = op._ctx.counters()
        assert c.n_table_items == c.n_valid_items > 5000
        op.close()
    for k, v in out['0'].items():
        if isinstance(v, np.ndarray):
            assert np.array_equal(out['2'][k], v, equal_nan=True), k
            assert np.array_equal(out['2+interp'][k], v, equal_nan=True), k
    assert np.isfinite(out['0']['RVEL']).sum() > 1000 and np.isfinite(out['0']['ZH']).sum() > 1000


def test_sweep_after_a_failed_sweep_equals_a_fresh_context(monkeypatch):
    """A launch sequence that returns an error AFTER its counting kernel was queued (a failed copy or capture; here:
    the test hook `fail_next_sweep`) must not leave its counter set half used: the sweep's two counter sets are used
    in turn and the serial advances only when a sequence is queued completely, so the next sweep would count on top
    of the stale bucket counts / rare-item totals (round-4 advisor finding; RadarOperator._simulate_sweeps retries on
    the same context after a MemoryError).  With the counting sort (the counts index the permutation) and with the
    default direct listing, one and nine sub-beams: the sweep after the failure has the bits and the counters of a
    sweep on a fresh context."""
    import bench
    from cosmo_pol_amd import RadarOperator, synthetic
    from cosmo_pol_amd._native import NativeError
    hyds = ('R', 'S', 'G')
    cube = synthetic.small_test_cube(hydrometeors=hyds)
    data = {k: v.copy() for k, v in cube['data'].items()}
    ny, nx = data['QR_v'].shape[1:]
    yy, xx = np.meshgrid(np.arange(ny), np.arange(nx), indexing='ij')
    wedge = (xx > nx // 2) & (yy > ny // 2)                     # items outside the tables: the rare-item totals matter too
    for k, tiny in (('QR_v', 1e-16), ('QS_v', 3e-18)):
        f = data[k]
        f[:, wedge] = np.where(f[:, wedge] > 0, np.float32(tiny) * (1 + (np.arange(f.shape[0]) % 5))[:, None], 0).astype(np.float32)
    luts = synthetic.make_all_luts(hyds, 5.6, '1mom', n_e=8)
    az = np.arange(20.0, 70.0, 2.5)
    el = np.full(len(az), 2.0)
    keys = ('ZH', 'ZV', 'ZDR', 'KDP', 'RHOHV', 'PHIDP', 'DELTA_HV', 'ATT_H', 'ATT_V', 'RVEL', 'mask')
    for n_gh in (1, 3):
        conf = bench.bench_config(True)
        conf['integration'].update(nh_GH=n_gh, nv_GH=n_gh)
        for env in ({'CPOL_RARE_DIRECT': '0', 'CPOL_GATE1': '0'}, {}):
            for k in ('CPOL_RARE_DIRECT', 'CPOL_GATE1'):
                monkeypatch.delenv(k, raising=False)
            for k, v in env.items():
                monkeypatch.setenv(k, v)
            op = RadarOperator(config=conf, luts=luts, output_variables='only_radar', lanes=1)
            op.load_model_arrays(data, cube['zlevels'], cube['proj_info'], cube['resolution'])
            ref = {k: v.copy() for k, v in op.simulate_rays(az, el).items() if k in keys}
            c0 = op._ctx.counters()
            want = (int(c0.n_valid_items), int(c0.n_table_items), int(c0.n_work_units))
            assert want[0] > want[1] > 0
            for failures in (1, 2):                             # (an odd and an even number of failed sequences)
                for _ in range(failures):
                    assert op._ctx.lib.cpol_debug_read(op._ctx.h, b'fail_next_sweep', None, 0) == 0
                    with pytest.raises(NativeError, match='fail_next_sweep'):
                        op.simulate_rays(az, el)
                got = op.simulate_rays(az, el)
                c = op._ctx.counters()
                assert (int(c.n_valid_items), int(c.n_table_items), int(c.n_work_units)) == want, (n_gh, env, failures)
                for k in keys:
                    assert np.array_equal(got[k], ref[k], equal_nan=True), (n_gh, env, failures, k)
            op.close()


def test_two_moment_items_outside_the_tables_in_place_integration(monkeypatch):
    """The two-launch single-beam sweep (k_interp_sweep + k_gate1_ray) integrates an item outside its integral table in
    place, lanes 0..7 of the wavefront standing in for the eight wavefronts of the integrating kernels.  Here for the
    2-moment gamma species (nu != 1: one exp per bin, k_psd<GAMMA_EXP>; rain, snow, graupel, hail without ice
    crystals), whose slope is bounded by the clip of the mean mass, so that nothing leaves a complete table: the test
    keeps only the lower panels of every table (CPOL_ITAB_KEEP_PANELS, as a table that lost panels to the accuracy gate)
    and forces the path (CPOL_GATE1_RAY=2).  Same bits as the sequence with the integrating launch and k_final
    (CPOL_GATE1_RAY=0) and as the general sequence with the counting sort; range scans and sensitivity cut included."""
    import bench
    from cosmo_pol_amd import RadarOperator, synthetic
    hyds = ('R', 'S', 'G', 'H')
    cube = synthetic.small_test_cube(hydrometeors=('R', 'S', 'G'), two_moment=True)
    conf = bench.bench_config(True)
    conf['microphysics'].update(scheme='2mom', with_ice_crystals=0)
    luts = synthetic.make_all_luts(hyds, 5.6, '2mom', n_e=8)
    az = np.arange(20.0, 70.0, 2.5)
    el = np.full(len(az), 2.0)
    # where do the items lie?  (all on the table by default)
    op = RadarOperator(config=conf, luts=luts, output_variables='only_radar', lanes=1)
    op.load_model_arrays(cube['data'], cube['zlevels'], cube['proj_info'], cube['resolution'])
    op.simulate_rays(az, el)
    c = op._ctx.counters()
    assert c.n_table_items == c.n_valid_items > 1000
    n_pan = [op._ctx.itab_detail(j)['n_pan'] for j in range(len(hyds))]
    op.close()
    out = {}
    monkeypatch.setenv('CPOL_ITAB_KEEP_PANELS', '0:%d' % (min(n_pan) * 5 // 8))
    for mode, env in (('sorted', {'CPOL_RARE_DIRECT': '0', 'CPOL_GATE1': '0'}), ('final', {'CPOL_GATE1_RAY': '0'}), ('ray', {'CPOL_GATE1_RAY': '2'})):
        for k in ('CPOL_RARE_DIRECT', 'CPOL_GATE1', 'CPOL_GATE1_RAY'):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        op = RadarOperator(config=conf, luts=luts, output_variables='only_radar', lanes=1)
        op.load_model_arrays(cube['data'], cube['zlevels'], cube['proj_info'], cube['resolution'])
        res = op.simulate_rays(az, el)
        c = op._ctx.counters()
        out[mode] = (res, int(c.n_valid_items), int(c.n_table_items))
        op.close()
    assert out['sorted'][1] == out['ray'][1] == out['final'][1] and out['sorted'][2] == out['ray'][2] == out['final'][2]
    assert out['ray'][1] - out['ray'][2] > 50 and out['ray'][2] > 50, out['ray'][1:]      # items integrated bin by bin, and items on the tables
    for k in ('ZH', 'ZV', 'ZDR', 'KDP', 'RHOHV', 'PHIDP', 'DELTA_HV', 'ATT_H', 'ATT_V', 'RVEL', 'mask'):
        assert np.array_equal(out['ray'][0][k], out['sorted'][0][k], equal_nan=True), k
        assert np.array_equal(out['final'][0][k], out['sorted'][0][k], equal_nan=True), k
    assert np.isfinite(out['ray'][0]['ZH']).sum() > 300


def test_rare_items_of_every_flavour_in_one_launch(monkeypatch):
    """k_psd_rare runs every integrating flavour -- melting (polynomial tables), 1-moment ice (lambda tables and the direct
    sums), the gamma recurrence -- from ONE launch when the work units are directly listed items.  A C3-like sweep (R, S, G,
    mS, mG, I; nine sub-beams) with a wedge of mass densities so small that items of every species leave their tables:
    same bits as one launch per flavour (CPOL_PSD_RARE=0) and as the counting sort (CPOL_RARE_DIRECT=0)."""
    import bench
    from cosmo_pol_amd import RadarOperator, synthetic
    hyds = ['R', 'S', 'G', 'mS', 'mG', 'I']
    conf = bench.bench_config(True, 'c3')
    conf['integration'].update(nh_GH=3, nv_GH=3)
    cube = synthetic.small_test_cube(hydrometeors=('R', 'S', 'G', 'I'))
    data = {k: v.copy() for k, v in cube['data'].items()}
    ny, nx = data['QR_v'].shape[1:]
    yy, xx = np.meshgrid(np.arange(ny), np.arange(nx), indexing='ij')
    wedge = (xx > nx // 2) & (yy > ny // 2)
    for k, tiny in (('QR_v', 1e-16), ('QS_v', 3e-18), ('QG_v', 1e-17), ('QI_v', 1e-19)):
        f = data[k]
        f[:, wedge] = np.where(f[:, wedge] > 0, np.float32(tiny), 0).astype(np.float32)
    luts = synthetic.make_all_luts(hyds, 5.6, '1mom', n_e=8)
    az = np.arange(20.0, 70.0, 2.5)
    el = np.full(len(az), 3.0)
    out = {}
    for mode, env in (('sorted', {'CPOL_RARE_DIRECT': '0'}), ('per_flavour', {'CPOL_PSD_RARE': '0'}), ('one_launch', {})):
        for k in ('CPOL_RARE_DIRECT', 'CPOL_PSD_RARE'):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        op = RadarOperator(config=conf, luts=luts, output_variables='only_radar', lanes=1)
        op.load_model_arrays(data, cube['zlevels'], cube['proj_info'], cube['resolution'])
        res = op.simulate_rays(az, el)
        c = op._ctx.counters()
        out[mode] = (res, int(c.n_valid_items), int(c.n_table_items))
        op.close()
    assert out['sorted'][1:] == out['one_launch'][1:] == out['per_flavour'][1:]
    assert out['one_launch'][1] - out['one_launch'][2] > 200, out['one_launch'][1:]
    for k in ('ZH', 'ZV', 'ZDR', 'KDP', 'RHOHV', 'PHIDP', 'DELTA_HV', 'ATT_H', 'ATT_V', 'RVEL', 'mask'):
        assert np.array_equal(out['one_launch'][0][k], out['sorted'][0][k], equal_nan=True), k
        assert np.array_equal(out['per_flavour'][0][k], out['sorted'][0][k], equal_nan=True), k
    assert np.isfinite(out['one_launch'][0]['ZH']).sum() > 300


def test_second_model_with_another_south_pole_rebuilds_the_coordinate_polynomials():
    """Round-5 advisor finding (medium): the coordinate polynomials of a single-beam sweep's resident table set hold the
    rotated-pole matrix of the model they were made for.  Staging a second cube with ANOTHER south pole and running the same
    rays again (the host's 'rays' cache and the library's table set survive the staging) must rebuild them: the default
    form then agrees with the long form of the geodesy (debug_flags = CPOL_DEBUG_EXACT_SUBBEAMS) on the new cube -- before
    the fix it read the grid cells of the old rotation (or left the domain) without any error."""
    import bench
    import torch
    from cosmo_pol_amd import RadarOperator, synthetic
    from cosmo_pol_amd import _native as N
    from cosmo_pol_oracle import geodesy
    conf = bench.bench_config(True)
    hyds = ('R', 'S', 'G')
    luts = synthetic.make_all_luts(hyds, 5.6, '1mom', n_e=8)
    site = conf['radar']['coords']
    poles = [(-43.0, 10.0), (-38.5, 13.0)]
    cubes = []
    for sp in poles:
        rot = geodesy.wgs_to_rotated(np.array([site[0]]), np.array([site[1]]), sp[0], sp[1])[0]      # (rotated lat, lon of the site)
        cubes.append(synthetic.small_test_cube(center_rot=(float(rot[0]), float(rot[1])), hydrometeors=hyds, south_pole=sp))
    assert cubes[0]['proj_info']['Lo1'] != cubes[1]['proj_info']['Lo1']
    op = RadarOperator(config=conf, luts=luts, output_variables='only_radar', lanes=1)
    az = np.arange(0.0, 360.0, 3.0)
    el = np.full(len(az), 1.5)
    ng = len(op.constants.RANGE_RADAR)
    slab = torch.empty((len(bench.RADAR_FIELDS), len(az), ng), dtype=torch.float32, device='cuda')
    ptrs = {k: slab[i].data_ptr() for i, k in enumerate(bench.RADAR_FIELDS)}
    results = []
    for cube in cubes:
        op.load_model_arrays(cube['data'], cube['zlevels'], cube['proj_info'], cube['resolution'])
        for _ in range(2):
            op.simulate_rays(az, el, device_outputs=ptrs)
        op.wait()
        assert op._ctx.launch_forms()['poly_central'] == 1
        poly = slab.cpu().numpy().copy()
        op.debug_flags = N.DEBUG_EXACT_SUBBEAMS
        op.simulate_rays(az, el, device_outputs=ptrs)
        op.wait()
        assert op._ctx.launch_forms()['poly_central'] == 0
        op.debug_flags = 0
        exact = slab.cpu().numpy().copy()
        assert np.isfinite(exact[0]).sum() > 2000
        assert np.array_equal(np.isnan(poly), np.isnan(exact))
        ok = np.isfinite(exact) & (exact != 0)
        assert np.max(np.abs(poly[ok] - exact[ok]) / np.abs(exact[ok])) < 1e-5
        results.append(exact)
    # (the two rotations put different model columns under the same rays: the test would not see a stale matrix otherwise)
    assert not np.array_equal(np.nan_to_num(results[0]), np.nan_to_num(results[1]))
    op.close()


def test_pipelined_single_beam_scan_drains_its_lanes_when_a_sweep_fails():
    """get_PPI of a single-beam scan queues a sweep per lane with page-locked outputs and waits once (round 6,
    RadarOperator.pipeline_single_beam_scans).  A sweep that fails on the way (the test hook `fail_next_sweep` on the lane
    that takes the second sweep) must surface as the call's exception with every lane drained -- no copy may still be
    writing into a block the caller never received -- and the same scan must then give the bits of the one launch sequence."""
    import bench
    from cosmo_pol_amd import RadarOperator, synthetic
    from cosmo_pol_amd._native import NativeError
    hyds = ('R', 'S', 'G')
    cube = synthetic.small_test_cube(hydrometeors=hyds)
    luts = synthetic.make_all_luts(hyds, 5.6, '1mom', n_e=8)
    conf = bench.bench_config(True)
    elevs = [1.0, 2.0, 4.0, 7.0]
    op = RadarOperator(config=conf, luts=luts, output_variables='only_radar', lanes=3)
    op.load_model_arrays(cube['data'], cube['zlevels'], cube['proj_info'], cube['resolution'])
    good = op.get_PPI(elevs, az_step=6.0)
    lane1 = op._lane(1)                                           # (the lane of the second sweep)
    assert lane1.lib.cpol_debug_read(lane1.h, b'fail_next_sweep', None, 0) == 0
    with pytest.raises(NativeError, match='fail_next_sweep'):
        op.get_PPI(elevs, az_step=6.0)
    for i in range(3):                                            # nothing left in flight on any lane
        assert op._lane(i).submitted == op._lane(i).completed, i
    again = op.get_PPI(elevs, az_step=6.0)
    op.pipeline_single_beam_scans = False
    one_seq = op.get_PPI(elevs, az_step=6.0)
    for i in range(len(elevs)):
        for name in good.fields:
            a = np.ma.asarray(good.get_field(i, name))
            for other in (again, one_seq):
                b = np.ma.asarray(other.get_field(i, name))
                assert np.array_equal(np.ma.getmaskarray(a), np.ma.getmaskarray(b)), name
                assert np.array_equal(a.filled(0), b.filled(0)), name
    op.close()
