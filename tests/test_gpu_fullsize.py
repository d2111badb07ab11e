"""Full-size (BASELINE configs[1]: 360 x 500 PPI on the 80 x 774 x 1158 bench cube)
GPU tests through size-independent properties + a ray sample against the oracle."""
import numpy as np
import pytest

import _cases
import bench
from cosmo_pol_amd import synthetic
from cosmo_pol_oracle import beam, scatter
from cosmo_pol_oracle import config as ocfg

pytestmark = pytest.mark.gpu
FIELDS = ['ZH', 'ZV', 'ZDR', 'KDP', 'DELTA_HV', 'PHIDP', 'RHOHV', 'ATT_H', 'ATT_V', 'RVEL']


@pytest.fixture(scope='module')
def full():
    from cosmo_pol_amd import RadarOperator
    conf = bench.bench_config(False)
    hyds = ('R', 'S', 'G')
    cube = synthetic.make_cube(hydrometeors=hyds, **synthetic.BENCH_GRID)
    luts = synthetic.make_all_luts(hyds, 5.6, '1mom')
    op = RadarOperator(config=conf, luts=luts, output_variables='only_radar')
    op.load_model_arrays(cube['data'], cube['zlevels'], cube['proj_info'], cube['resolution'])
    az = np.arange(0, 360, 1.0)
    res = op.simulate_rays(az, np.full(360, 1.0))
    yield dict(op=op, conf=conf, cube=cube, luts=luts, az=az, res=res)
    op.close()


def test_full_ppi_shapes_and_counters(full):
    res, op = full['res'], full['op']
    assert res['ZH'].shape == (360, 500)
    cnt = op._ctx.counters()
    assert cnt.n_subbeam_gates == 180000 and cnt.n_gates == 180000
    assert 100000 < cnt.n_valid_items < 540000
    assert np.isfinite(res['ZH']).mean() > 0.5
    assert np.all(res['ZH'][np.isfinite(res['ZH'])] > 0)
    assert np.nanmax(res['RHOHV']) <= 1.0 + 1e-6


def test_rays_are_independent_bitwise(full):
    """A radial's result does not depend on which other radials share its launch
    (batch of 360, batch of 7 in shuffled order, single ray): bitwise equal --
    this is what makes azimuth sharding over GPUs exact."""
    op, az, res = full['op'], full['az'], full['res']
    pick = np.array([300, 0, 45, 179, 180, 259, 12])
    sub = op.simulate_rays(az[pick], np.full(len(pick), 1.0))
    one = op.simulate_rays(az[[45]], np.array([1.0]))
    for k in FIELDS + ['mask', 'lats', 'lons', 'dist', 'heights']:
        assert np.array_equal(sub[k], res[k][pick], equal_nan=True), k
        assert np.array_equal(one[k][0], res[k][45], equal_nan=True), k
    again = op.simulate_rays(az, np.full(360, 1.0))
    for k in FIELDS:
        assert np.array_equal(again[k], res[k], equal_nan=True), 'run-to-run ' + k


def test_sample_of_rays_vs_oracle(full):
    from test_gpu_parity import _pol_tolerances
    conf = ocfg.make_config(full['conf'])
    cube, luts, res = full['cube'], full['luts'], full['res']
    oc = beam.ModelCube({n: cube['data'][n] for n in _cases.ORDER}, cube['zlevels'],
                        cube['proj_info'], cube['resolution'], _cases.ORDER)
    ol = {h: _cases.as_oracle_lut(l) for h, l in luts.items()}
    n_valid = 0
    for r in (0, 37, 90, 123, 181, 222, 275, 301, 359):
        subs = beam.interpolate_radial(oc, conf, float(full['az'][r]), 1.0)
        o = scatter.radar_observables(subs, ol, conf, return_sz=True)
        szt = np.nan_to_num(o.sz_total.astype(np.float64))
        scatter.cut_at_sensitivity([o], conf)
        assert np.array_equal(res['dist'][r], subs[0].dist_profile)
        assert np.array_equal(res['heights'][r], subs[0].heights_profile)
        assert np.array_equal(res['mask'][r], o.mask)
        for k in FIELDS:
            atol = 2e-4 if k == 'RVEL' else _pol_tolerances(k, o, szt, conf)
            _cases.assert_close_nan(res[k][r], o.values[k], rtol=1e-5, atol=atol,
                                    name='%s ray %d' % (k, r))
        n_valid += int(np.isfinite(o.values['ZH']).sum())
    assert n_valid > 2000


def test_c3_full_size_volume_vs_oracle():
    """BASELINE configs[2] at full size: 5 elevations x (360 x 500) on the bench cube with
    the full 1-moment hydrometeor set (R, S, G, melting snow / graupel, ice crystals).
    Size-independent properties (lanes == sequential is tested below; rays are independent)
    plus sampled rays of every elevation against the oracle."""
    from cosmo_pol_amd import RadarOperator
    from test_gpu_parity import _pol_tolerances
    over = bench.bench_config(False)
    over['microphysics'].update(with_melting=1, with_ice_crystals=1)
    hyds = ['R', 'S', 'G', 'mS', 'mG', 'I']
    cube = synthetic.make_cube(hydrometeors=('R', 'S', 'G', 'I'), **synthetic.BENCH_GRID)
    luts = synthetic.make_all_luts(hyds, 5.6, '1mom')
    op = RadarOperator(config=over, luts=luts, output_variables='only_radar', lanes=3)
    op.load_model_arrays(cube['data'], cube['zlevels'], cube['proj_info'], cube['resolution'])
    elevs = [0.5, 1.5, 3.0, 5.0, 8.0]
    scan = op.get_PPI(elevs, az_step=1.0)
    assert scan.nsweeps == 5 and scan.fields['ZH']['data'].shape == (5 * 360, 500)
    conf = ocfg.make_config(over)
    oc = beam.ModelCube({n: cube['data'][n] for n in _cases.ORDER}, cube['zlevels'],
                        cube['proj_info'], cube['resolution'], _cases.ORDER)
    ol = {h: _cases.as_oracle_lut(l) for h, l in luts.items()}
    n_valid = n_melt = 0
    for i, e in enumerate(elevs):
        raw = scan.raw[i]
        for r in ((17 + 71 * i) % 360, (200 + 53 * i) % 360):
            subs = beam.interpolate_radial(oc, conf, float(r), e)
            n_melt += int(np.sum(np.asarray(subs[0].values['QmS_v']) > 0)) if 'QmS_v' in subs[0].values else 0
            o = scatter.radar_observables(subs, ol, conf, return_sz=True)
            szt = np.nan_to_num(o.sz_total.astype(np.float64))
            scatter.cut_at_sensitivity([[o]], conf)
            assert np.array_equal(raw['dist'][r], subs[0].dist_profile)
            assert np.array_equal(raw['heights'][r], subs[0].heights_profile)
            assert np.array_equal(raw['mask'][r], o.mask)
            for k in FIELDS:
                atol = 2e-4 if k == 'RVEL' else _pol_tolerances(k, o, szt, conf)
                _cases.assert_close_nan(raw['fields'][k][r], o.values[k], rtol=1e-5, atol=atol,
                                        name='%s el %g ray %d' % (k, e, r))
            n_valid += int(np.isfinite(o.values['ZH']).sum())
    assert n_valid > 2000 and n_melt > 20, (n_valid, n_melt)
    op.close()


def test_c4_sector_with_49_subbeams_vs_oracle(monkeypatch):
    """BASELINE configs[3] on a sector: 45 azimuths x 500 gates x 7 x 7 sub-beams at two elevations
    on the bench cube, full 1-moment set with melting.  This is the path of the multi-GPU
    workload -- the melting items grouped by table block over tiles of 16 rays x 4 gates
    (k_psd_lookup), the items on 1-D tables evaluated inside the sub-beam accumulation
    (k_subbeam_sum), the velocity terms in their own kernel (k_rvel_terms) -- against the oracle
    on sampled rays; the same rays alone (one ray per call: no tiles) must give the same bits, and so
    must the form of k_subbeam_sum that takes the coefficient rows of a tile's distinct table blocks
    through the scalar cache (CPOL_SUBSUM_COOP=1; by default only launches of >= 16 wavefronts per SIMD
    use it) against the per-lane gather (CPOL_SUBSUM_COOP=0)."""
    from cosmo_pol_amd import RadarOperator
    from test_gpu_parity import _pol_tolerances
    over = bench.bench_config(False, 'c4')
    hyds = list(bench.hydrometeors_of('c4'))
    cube = synthetic.make_cube(hydrometeors=('R', 'S', 'G', 'I'), **synthetic.BENCH_GRID)
    luts = synthetic.make_all_luts(hyds, 5.6, '1mom')
    op = RadarOperator(config=over, luts=luts, output_variables='only_radar', lanes=1)
    op.load_model_arrays(cube['data'], cube['zlevels'], cube['proj_info'], cube['resolution'])
    conf = ocfg.make_config(over)
    oc = beam.ModelCube({n: cube['data'][n] for n in _cases.ORDER}, cube['zlevels'],
                        cube['proj_info'], cube['resolution'], _cases.ORDER)
    ol = {h: _cases.as_oracle_lut(l) for h, l in luts.items()}
    az = np.arange(100.0, 145.0, 1.0)       # 45 rays: the last tile of 16 rays is partly empty
    n_melt = n_valid = 0
    for e in (1.5, 5.0):
        res = op.simulate_rays(az, np.full(len(az), e))
        c = op._ctx.counters()
        assert res['n_sub'] == 49 and c.n_table_items >= c.n_valid_items - 8
        for r in (5, 30):
            subs = beam.interpolate_radial(oc, conf, float(az[r]), e)
            assert len(subs) == 49
            n_melt += sum(int(np.sum(np.asarray(sb.values['QmS_v']) > 0)) for sb in subs if 'QmS_v' in sb.values)
            o = scatter.radar_observables(subs, ol, conf, return_sz=True)
            szt = np.nan_to_num(o.sz_total.astype(np.float64))
            scatter.cut_at_sensitivity([[o]], conf)
            assert np.array_equal(res['mask'][r], o.mask)
            for k in FIELDS:
                atol = 2e-4 if k == 'RVEL' else _pol_tolerances(k, o, szt, conf)
                _cases.assert_close_nan(res[k][r], o.values[k], rtol=1e-5, atol=atol,
                                        name='%s el %g ray %d' % (k, e, r))
            n_valid += int(np.isfinite(o.values['ZH']).sum())
            alone = op.simulate_rays(az[r:r + 1], np.full(1, e))
            for k in FIELDS:
                assert np.array_equal(alone[k][0], res[k][r], equal_nan=True), (k, e, r)
    assert n_valid > 300 and n_melt > 100, (n_valid, n_melt)
    op.close()
    forms = {}
    for form in ('gather', 'gather1', 'coop', 'scalar', 'tail', 'two_kernels', 'team4', 'team6', 'team7', 'chain2', 'chain4', 'chain8', 'listed_tiles'):
        # 'gather': the per-lane gather in its small-launch form (three wavefronts per (tile, hydrometeor), the
        # whole block requested at once: CPOL_SUBSUM_SMALL=1, an experiment); 'gather1': one wavefront per
        # (tile, hydrometeor), rows two at a time (what a small launch gets by default);
        # 'coop': up to 6 table blocks per wavefront and sub-beam staged in LDS (global_load_lds, round 4), the
        # remaining lanes by the gather tail; 'scalar': the same walk with the rows through the scalar cache
        # (round 3's form, CPOL_SUBSUM_FORM=scalar); 'tail': ONE block that way, every other lane through the tail
        # 'team<W>': W wavefronts per (tile, species), the r-th present sub-beam evaluated by wavefront r mod W, the terms
        # of a round handed through LDS and added in order (k_subbeam_sum_team, round 5);
        # 'chain<W>': the team with the tile's float32 sums waiting in LDS, handed from sub-beam to sub-beam (CPOL_SUBSUM_CHAIN=1: no
        # barrier per round);
        # 'listed_tiles': the workgroups of k_psd_lookup list the tiles with a melting species among their own and walk that list (by
        # default only from 262 144 tiles on: CPOL_LOOKUP_LIST=2 asks for it here) instead of one wavefront per tile;
        # 'two_kernels': the default forms, but k_interp_sweep + k_classify instead of the one kernel that interpolates
        # and classifies its gates (k_interp_classify, the default of this path since round 4)
        monkeypatch.setenv('CPOL_SUBSUM_COOP', '0' if form.startswith('gather') else '1')     # (read when the context is created)
        monkeypatch.setenv('CPOL_SUBSUM_SMALL', '0' if form == 'gather1' else '1')
        monkeypatch.setenv('CPOL_SUBSUM_COOP_ROUNDS', '1' if form == 'tail' else '6')
        monkeypatch.setenv('CPOL_SUBSUM_FORM', 'scalar' if form == 'scalar' else 'lds')
        monkeypatch.setenv('CPOL_FUSE_CLASSIFY', '0' if form == 'two_kernels' else '1')
        monkeypatch.setenv('CPOL_SUBSUM_TEAM', form[4:] if form.startswith('team') else form[5:] if form.startswith('chain') else '0')
        monkeypatch.setenv('CPOL_SUBSUM_CHAIN', '1' if form.startswith('chain') else '0')
        monkeypatch.setenv('CPOL_LOOKUP_LIST', '2' if form == 'listed_tiles' else '1')
        if form == 'two_kernels':
            monkeypatch.delenv('CPOL_SUBSUM_COOP'); monkeypatch.delenv('CPOL_SUBSUM_SMALL')
        opc = RadarOperator(config=over, luts=luts, output_variables='only_radar', lanes=1)
        opc.load_model_arrays(cube['data'], cube['zlevels'], cube['proj_info'], cube['resolution'])
        forms[form] = opc.simulate_rays(az, np.full(len(az), 5.0))
        assert opc._ctx.counters().n_table_items > 100000
        opc.close()
    for k in FIELDS:
        assert np.array_equal(forms['gather'][k], forms['coop'][k], equal_nan=True), k
        assert np.array_equal(forms['gather'][k], forms['tail'][k], equal_nan=True), k
        assert np.array_equal(forms['gather'][k], forms['gather1'][k], equal_nan=True), k
        assert np.array_equal(forms['gather'][k], forms['scalar'][k], equal_nan=True), k
        assert np.array_equal(forms['gather'][k], forms['two_kernels'][k], equal_nan=True), k
        for w in (4, 6, 7):
            assert np.array_equal(forms['gather'][k], forms['team%d' % w][k], equal_nan=True), (k, w)
        for w in (2, 4, 8):
            assert np.array_equal(forms['gather'][k], forms['chain%d' % w][k], equal_nan=True), (k, w)
        assert np.array_equal(forms['gather'][k], forms['listed_tiles'][k], equal_nan=True), k
        assert np.array_equal(forms['gather'][k], res[k], equal_nan=True), k     # (res: the default choice, last elevation)


def test_rhi_and_vprof_api(full):
    op = full['op']
    rhi = op.get_RHI(azimuths=[30.0, 200.0], elevations=np.arange(0.5, 20.0, 2.5))
    op.volume_in_one_sequence = False                      # sweep by sweep (lanes): the same bits
    rhi2 = op.get_RHI(azimuths=[30.0, 200.0], elevations=np.arange(0.5, 20.0, 2.5))
    op.volume_in_one_sequence = True
    for k in ('ZH', 'ZDR', 'KDP', 'RVEL'):
        assert np.array_equal(np.ma.getdata(rhi.fields[k]['data']), np.ma.getdata(rhi2.fields[k]['data']), equal_nan=True), k
    assert rhi.nsweeps == 2 and rhi.scan_type == 'rhi'
    assert rhi.fields['ZH']['data'].shape == (16, 500)
    assert np.allclose(rhi.fixed_angle['data'], [30.0, 200.0])
    assert np.array_equal(rhi.sweep_start_ray_index['data'], [0, 8])
    z = rhi.get_field(1, 'ZDR')
    assert z.shape == (8, 500)
    vp = op.get_VPROF()
    assert vp.fields['KDP']['data'].shape == (1, 500)
    # vertical beam: first gates are rain, then snow -> the profile leaves the model top
    assert np.ma.count(vp.fields['ZH']['data']) > 20


def test_lanes_volume_scan_equals_sequential():
    """A volume scan four ways -- its sweeps one after the other, spread over forked contexts
    (cpol_fork, one host thread per lane), as ONE launch sequence (get_PPI's form for scans with sub-beams: rays
    of all elevations in one cpol_run_sweep call), and queued sweep by sweep on the lanes with page-locked outputs and
    one wait at the end (get_PPI's form for single-beam scans, round 6) -- gives bit-identical fields."""
    from cosmo_pol_amd import RadarOperator, synthetic
    import bench
    conf = bench.bench_config(True)
    conf['microphysics'].update(with_melting=1, with_ice_crystals=1)
    hyds = ['R', 'S', 'G', 'mS', 'mG', 'I']
    cube = synthetic.small_test_cube(hydrometeors=('R', 'S', 'G', 'I'))
    luts = synthetic.make_all_luts(hyds, 5.6, '1mom', n_e=8)
    elevs = [0.5, 1.5, 3.0, 5.0, 8.0]
    scans = []
    for lanes, one_sequence, pipelined in ((1, False, False), (3, False, False), (3, True, False), (2, 'budget', False), (3, True, True)):
        op = RadarOperator(config=conf, luts=luts, output_variables='all', lanes=lanes)
        assert op.pipeline_single_beam_scans is True          # (the default)
        op.volume_in_one_sequence = bool(one_sequence)
        op.pipeline_single_beam_scans = pipelined
        op.load_model_arrays(cube['data'], cube['zlevels'], cube['proj_info'], cube['resolution'])
        if one_sequence == 'budget':
            # a memory budget of two sweeps' work buffers: the scan runs as sequences of 2 + 2 + 1 sweeps
            free, total, per_gate = op._ctx.mem_info()
            assert 0 < free <= total and 500 < per_gate < 4000
            op.sequence_memory_budget = int(2.5 * per_gate * 90 * len(op.constants.RANGE_RADAR))
            groups = op._sweep_groups([(np.arange(0, 360, 4.0), np.full(90, e)) for e in elevs])
            assert [len(g) for g in groups] == [2, 2, 1]
        scans.append(op.get_PPI(elevs, az_step=4.0))
        if lanes == 3 and not one_sequence:
            assert len(op._lane_ctx) == 2
            # staging on a context with live lanes is refused by the library; the operator
            # drops its lanes first
            with pytest.raises(ValueError):
                op._ctx.set_num_hydro(len(hyds))
            op.set_lut()
            assert op._lane_ctx == []
        if one_sequence and not pipelined:
            assert op._lane_ctx == []                    # calls on the root context only
        if pipelined:
            assert len(op._lane_ctx) == 2                # lanes 0 (the root), 1, 2 took the five sweeps in turn
        op.close()
    a = scans[0]
    for b in scans[1:]:
        for i in range(len(elevs)):
            for name in a.fields:
                x, y = np.ma.asarray(a.get_field(i, name)), np.ma.asarray(b.get_field(i, name))
                assert np.array_equal(np.ma.getmaskarray(x), np.ma.getmaskarray(y)), name
                assert np.array_equal(x.filled(0), y.filled(0)), name
    # results stay valid after the operators are closed (they live in pooled pinned blocks of their own)
    assert np.isfinite(scans[2].raw[2]['fields']['ZH']).sum() > 100


def test_graph_replay_equals_plain_launches(monkeypatch):
    """Device-output sweeps are captured into a HIP graph and replayed while nothing changes;
    the result must equal the plain launch sequence (CPOL_USE_GRAPH=0), also after the scan
    geometry changed and changed back."""
    import torch
    from cosmo_pol_amd import RadarOperator, synthetic
    conf = bench.bench_config(True)
    hyds = ('R', 'S', 'G')
    cube = synthetic.small_test_cube(hydrometeors=hyds)
    luts = synthetic.make_all_luts(hyds, 5.6, '1mom', n_e=8)
    az_a, az_b = np.arange(0., 360., 3.), np.arange(1., 361., 3.)
    el = np.full(len(az_a), 2.0)
    fields = ['ZH', 'ZDR', 'KDP', 'RHOHV', 'PHIDP']
    results = {}
    for use_graph in ('1', '0'):
        monkeypatch.setenv('CPOL_USE_GRAPH', use_graph)
        op = RadarOperator(config=conf, luts=luts, output_variables='only_radar', lanes=1)
        op.load_model_arrays(cube['data'], cube['zlevels'], cube['proj_info'], cube['resolution'])
        n_gates = len(op.constants.RANGE_RADAR)
        slab = torch.full((len(fields), len(az_a), n_gates), -1.0, dtype=torch.float32, device='cuda')
        ptrs = {k: slab[i].data_ptr() for i, k in enumerate(fields)}
        snaps = []
        for az in (az_a, az_a, az_a, az_b, az_b, az_a, az_a):
            op.simulate_rays(az, el, device_outputs=ptrs)
            op._ctx.synchronize()
            snaps.append(slab.cpu().numpy().copy())
        results[use_graph] = snaps
        op.close()
    for x, y in zip(results['1'], results['0']):
        assert np.array_equal(x, y, equal_nan=True)
    assert np.array_equal(results['1'][0], results['1'][2], equal_nan=True)
    assert not np.array_equal(results['1'][0], results['1'][3], equal_nan=True)


def test_c4_volume_in_one_launch_sequence_90_azimuths_all_elevations_vs_oracle():
    """BASELINE configs[3] as the multi-GPU bench runs it: the five elevations of the volume in ONE
    launch sequence (rays of different elevations in one cpol_run_sweep call), 90 azimuths x 5
    elevations x 49 sub-beams x 500 gates on the bench cube -- what two ranks of eight compute.
    Two rays of every elevation against the oracle (rays through the melting layer included), and
    the whole block bit-identical to the same rays computed sweep by sweep."""
    from cosmo_pol_amd import RadarOperator
    from test_gpu_parity import _pol_tolerances
    over = bench.bench_config(False, 'c4')
    hyds = list(bench.hydrometeors_of('c4'))
    cube = synthetic.make_cube(hydrometeors=('R', 'S', 'G', 'I'), **synthetic.BENCH_GRID)
    luts = synthetic.make_all_luts(hyds, 5.6, '1mom')
    op = RadarOperator(config=over, luts=luts, output_variables='only_radar', lanes=1)
    op.load_model_arrays(cube['data'], cube['zlevels'], cube['proj_info'], cube['resolution'])
    conf = ocfg.make_config(over)
    oc = beam.ModelCube({n: cube['data'][n] for n in _cases.ORDER}, cube['zlevels'],
                        cube['proj_info'], cube['resolution'], _cases.ORDER)
    ol = {h: _cases.as_oracle_lut(l) for h, l in luts.items()}
    az1 = np.arange(0.0, 360.0, 4.0)                                    # 90 azimuths
    elevs = bench.C4_ELEVATIONS
    az = np.tile(az1, len(elevs))
    el = np.repeat(np.asarray(elevs, dtype=float), len(az1))
    res = op.simulate_rays(az, el)
    c = op._ctx.counters()
    assert res['ZH'].shape == (450, 500) and res['n_sub'] == 49
    assert c.n_subbeam_gates == 450 * 49 * 500 and c.n_table_items >= c.n_valid_items - 64
    n_melt = n_valid = 0
    for i, e in enumerate(elevs):
        for r1 in ((7 + 19 * i) % 90, (52 + 11 * i) % 90):
            r = i * 90 + r1
            subs = beam.interpolate_radial(oc, conf, float(az[r]), float(e))
            n_melt += sum(int(np.sum(np.asarray(sb.values['QmS_v']) > 0)) for sb in subs if 'QmS_v' in sb.values)
            o = scatter.radar_observables(subs, ol, conf, return_sz=True)
            szt = np.nan_to_num(o.sz_total.astype(np.float64))
            scatter.cut_at_sensitivity([[o]], conf)
            assert np.array_equal(res['mask'][r], o.mask)
            assert np.array_equal(res['heights'][r], subs[24].heights_profile)
            for k in FIELDS:
                atol = 2e-4 if k == 'RVEL' else _pol_tolerances(k, o, szt, conf)
                _cases.assert_close_nan(res[k][r], o.values[k], rtol=1e-5, atol=atol,
                                        name='%s el %g az %g' % (k, e, az[r]))
            n_valid += int(np.isfinite(o.values['ZH']).sum())
    assert n_valid > 1500 and n_melt > 500, (n_valid, n_melt)
    # sweep by sweep (what a single-GPU get_PPI does): the same bits
    for i, e in enumerate(elevs):
        one = op.simulate_rays(az1, np.full(len(az1), float(e)))
        for k in FIELDS + ['mask', 'lats', 'lons', 'dist', 'heights']:
            assert np.array_equal(one[k], res[k][i * 90:(i + 1) * 90], equal_nan=True), (k, e)
    op.close()


def test_c4_full_volume_equals_the_shares_of_eight_ranks_bitwise():
    """BASELINE configs[3] at its FULL size -- 5 elevations x 360 azimuths x 49 sub-beams x 500 gates,
    44.1 M sub-beam gates in one launch sequence -- against the same volume computed as the eight
    contiguous 45-azimuth shares that eight ranks would compute (each share one launch sequence of
    its rays of all five sweeps): every field bit for bit.  The whole volume is large enough for the
    cooperative (LDS) form of k_subbeam_sum (16 wavefronts per SIMD and more), the shares take the per-lane
    gather, so this is also that pair at full size; rays are independent, so nothing else may differ."""
    from cosmo_pol_amd import RadarOperator
    over = bench.bench_config(False, 'c4')
    hyds = list(bench.hydrometeors_of('c4'))
    cube = synthetic.make_cube(hydrometeors=('R', 'S', 'G', 'I'), **synthetic.BENCH_GRID)
    luts = synthetic.make_all_luts(hyds, 5.6, '1mom')
    op = RadarOperator(config=over, luts=luts, output_variables='only_radar', lanes=1)
    op.load_model_arrays(cube['data'], cube['zlevels'], cube['proj_info'], cube['resolution'])
    elevs = np.asarray(bench.C4_ELEVATIONS, dtype=float)
    az1 = np.arange(0.0, 360.0, 1.0)
    full = op.simulate_rays(np.tile(az1, len(elevs)), np.repeat(elevs, len(az1)))
    c = op._ctx.counters()
    assert full['ZH'].shape == (1800, 500) and c.n_subbeam_gates == 1800 * 49 * 500
    assert c.n_valid_items > 50e6 and c.n_table_items >= c.n_valid_items - 64
    names = FIELDS + ['mask', 'lats', 'lons', 'dist', 'heights']
    full = {k: np.array(full[k], copy=True) for k in names}
    n_finite = 0
    for b in range(8):
        rays = az1[45 * b:45 * (b + 1)]
        part = op.simulate_rays(np.tile(rays, len(elevs)), np.repeat(elevs, len(rays)))
        assert op._ctx.counters().n_subbeam_gates == 225 * 49 * 500
        for s in range(len(elevs)):
            rows_full = slice(360 * s + 45 * b, 360 * s + 45 * (b + 1))
            rows_part = slice(45 * s, 45 * (s + 1))
            for k in names:
                assert np.array_equal(full[k][rows_full], part[k][rows_part], equal_nan=True), (k, b, s)
        n_finite += int(np.isfinite(part['ZH']).sum())
    assert n_finite > 100000
    op.close()


def test_short_form_of_the_subbeam_geodesy_keeps_every_cell_index():
    """The non-central sub-beams take a SHORT form of the geodesy (4 Vincenty passes, Newton reciprocal roots, short
    atan / asin series: cpol_interp.inl, CPOL_INTERP_FAST_SUB); cpol_sweep_params.debug_flags =
    CPOL_DEBUG_EXACT_SUBBEAMS sends them through the central sub-beam's long form.  On the C4 sector (45 rays x 49
    sub-beams x 500 gates, two elevations: 2.2 M sub-beam gates, 4.4 M float32 coordinates) the two forms must agree
    on every model cell (i0, i1) and every mask -- north_star's "bit-exact for gate/bin indexing" -- and only a few
    float32 coordinates in a million may differ, by a few ulp; the record of the full volume (44.1 M sub-beam gates:
    43 of 88.2 M coordinates differ, by 5 ulp at most, no cell, no mask, no model value, no output) is
    profiles/r5_fast_sub_check.json (tools/fast_sub_check.py)."""
    import importlib.util
    import os
    from cosmo_pol_amd import RadarOperator
    spec = importlib.util.spec_from_file_location(
        'fast_sub_check', os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tools', 'fast_sub_check.py'))
    fsc = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fsc)
    over = bench.bench_config(False, 'c4')
    hyds = list(bench.hydrometeors_of('c4'))
    cube = synthetic.make_cube(hydrometeors=('R', 'S', 'G', 'I'), **synthetic.BENCH_GRID)
    luts = synthetic.make_all_luts(hyds, 5.6, '1mom')
    op = RadarOperator(config=over, luts=luts, output_variables='only_radar', lanes=1)
    op.load_model_arrays(cube['data'], cube['zlevels'], cube['proj_info'], cube['resolution'])
    az = np.arange(100.0, 145.0, 1.0)
    n = 0
    for e in (1.5, 5.0):
        r = fsc.compare_sweep(op, az, np.full(len(az), e))
        assert r['n_subbeam_gates'] == 45 * 49 * 500
        assert r['b_cells_that_differ'] == 0 and r['b_masks_that_differ'] == 0 and r['c_nan_pattern_differs'] == 0, r
        assert r['a_max_ulp'] <= 8 and r['a_coordinates_that_differ'] <= 1e-5 * r['n_coordinates'], r
        assert r['radial_mask_equal']
        assert max(r['c_worst_relative_change_of_an_output'].values()) < 1e-5, r
        n += r['n_coordinates']
    assert n > 4.0e6
    op.close()


def test_single_beam_sweep_on_the_coordinate_polynomials_keeps_every_cell_index(monkeypatch):
    """A single-beam sweep whose float64 latitude / longitude nobody asks for (device outputs of the radar fields: bench.py's
    c2 / c3 steps) takes the rotated coordinates of its ONE sub-beam from the polynomials of its table set too (round 5), behind
    a guard: a gate whose float32 coordinate has a neighbour (1-2 ulp) in another model cell or outside the domain takes the
    long form after all -- the cell indices, the domain check and the masks are the long form's BY CONSTRUCTION, only the
    weights inside a cell may move by an ulp of the cell coordinate in a few gates per ten million.  Here: 8 sweeps of
    360 x 500 gates both ways (CPOL_GEO_POLY_CENTRAL=2 keeps the polynomials under the debug reads; debug_flags =
    CPOL_DEBUG_EXACT_SUBBEAMS is the long form); the record of 40 sweeps (14.4 M coordinates: 6 differ, 0 cells, 0 masks,
    0 model values, 0 outputs) is profiles/r5_central_poly_check.json (tools/fast_sub_check.py --config c3)."""
    import importlib.util
    import os
    from cosmo_pol_amd import RadarOperator
    monkeypatch.setenv('CPOL_GEO_POLY_CENTRAL', '2')
    spec = importlib.util.spec_from_file_location(
        'fast_sub_check', os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tools', 'fast_sub_check.py'))
    fsc = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fsc)
    fsc.DEVICE_OUTPUTS = True
    over = bench.bench_config(False, 'c3')
    hyds = list(bench.hydrometeors_of('c3'))
    cube = synthetic.make_cube(hydrometeors=('R', 'S', 'G', 'I'), **synthetic.BENCH_GRID)
    luts = synthetic.make_all_luts(hyds, 5.6, '1mom')
    op = RadarOperator(config=over, luts=luts, output_variables='only_radar', lanes=1)
    op.load_model_arrays(cube['data'], cube['zlevels'], cube['proj_info'], cube['resolution'])
    az = np.arange(0.0, 360.0, 1.0)
    n = 0
    for e in (0.5, 1.0, 2.0, 3.0, 5.0, 8.0, 12.0, 20.0):
        r = fsc.compare_sweep(op, az, np.full(len(az), e))
        assert r['n_subbeam_gates'] == 360 * 500
        assert r['b_cells_that_differ'] == 0 and r['b_masks_that_differ'] == 0 and r['c_nan_pattern_differs'] == 0, r
        assert r['a_max_ulp'] <= 8 and r['a_coordinates_that_differ'] <= 1e-5 * r['n_coordinates'], r
        assert max(r['c_worst_relative_change_of_an_output'].values()) < 1e-5, r
        n += r['n_coordinates']
    assert n == 8 * 360 * 500 * 2
    # ... and the polynomials WERE what the default form took (the last sweep of compare_sweep is the long form: run one more)
    import torch
    ng = len(op.constants.RANGE_RADAR)
    slab = torch.empty((len(bench.RADAR_FIELDS), len(az), ng), dtype=torch.float32, device='cuda')
    ptrs = {k: slab[i].data_ptr() for i, k in enumerate(bench.RADAR_FIELDS)}
    op.simulate_rays(az, np.full(len(az), 3.0), device_outputs=ptrs)
    op.wait()
    assert int(op._ctx.debug_read('poly_central', (1,), np.int32)[0]) == 1
    # host outputs ask for latitude / longitude: since round 6 the grid coordinates STILL come from the guarded polynomials (the
    # long form runs for the two float64 arrays alone), so that identical calls give identical bits whether or not the caller
    # fetches the gate coordinates (round-5 advisor finding); the float64 arrays are the long form's bit for bit
    res = op.simulate_rays(az[:8], np.full(8, 3.0))
    assert int(op._ctx.debug_read('poly_central', (1,), np.int32)[0]) == 1
    assert np.isfinite(res['lats']).all() and np.isfinite(res['lons']).all()
    from cosmo_pol_amd import _native as N
    op.debug_flags = N.DEBUG_EXACT_SUBBEAMS
    for k in [k for k in op._cache if isinstance(k, tuple) and k[0] == 'geom']:
        del op._cache[k]
    long_form = op.simulate_rays(az[:8], np.full(8, 3.0))
    assert int(op._ctx.debug_read('poly_central', (1,), np.int32)[0]) == 0
    op.debug_flags = 0
    for k in ('lats', 'lons', 'dist', 'heights'):
        assert np.array_equal(res[k], long_form[k], equal_nan=True), k
    # ... and a repeat of the first call (its gate coordinates now come from the host cache) gives the first call's bits
    again = op.simulate_rays(az[:8], np.full(8, 3.0))
    for k in bench.RADAR_FIELDS + ['RVEL', 'mask']:
        assert np.array_equal(res[k], again[k], equal_nan=True), k
    op.close()
