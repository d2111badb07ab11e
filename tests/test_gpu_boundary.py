"""GPU tests of the C-ABI boundary behaviour added in round 2: the sticky (deferred) domain
error, non-blocking pinned-host outputs, the re-validation in cpol_stage_hydro /
cpol_run_sweep, and the sensitivity cut of the Doppler spectrum through get_PPI."""
import copy
import os

import numpy as np
import pytest

import _cases
from cosmo_pol_oracle import beam, scatter
from cosmo_pol_oracle import config as ocfg

pytestmark = pytest.mark.gpu
FIELDS = ['ZH', 'ZV', 'ZDR', 'KDP', 'DELTA_HV', 'PHIDP', 'RHOHV', 'ATT_H', 'ATT_V', 'RVEL', 'mask',
          'lats', 'lons', 'dist', 'heights']


def _op(name, over_extra=None, output_variables='only_radar', lanes=1):
    from cosmo_pol_amd import RadarOperator
    over = copy.deepcopy(_cases.gen_golden.radial_case_inputs(name)[0])
    for sec, d in (over_extra or {}).items():
        over.setdefault(sec, {}).update(d)
    _, _, _, ocube, luts, cube = _cases.radial_case(name)
    op = RadarOperator(config=over, luts=luts, output_variables=output_variables, lanes=lanes)
    op.load_model_arrays(cube['data'], cube['zlevels'], cube['proj_info'], cube['resolution'])
    return op, over, ocube, luts


def test_deferred_domain_error_is_sticky_and_reported_once():
    """A sweep with device outputs returns before its kernels ran; if it left the model domain
    the error must survive later (clean) sweeps on the same context and surface exactly once
    (the reference raises IndexError at that radial, interpolation.py:572-580)."""
    import torch
    op, over, _, _ = _op('c2_rsg')
    n_gates = len(op.constants.RANGE_RADAR)
    slab = torch.zeros((2, 4, n_gates), dtype=torch.float32, device='cuda')
    ptrs = {'ZH': slab[0].data_ptr(), 'ZDR': slab[1].data_ptr()}
    az = np.array([0., 90., 180., 270.])
    op.simulate_rays(az, np.full(4, 2.0), device_outputs=ptrs)
    op.wait()                                               # clean so far
    far = copy.deepcopy(over)
    far['radar']['range'] = 150000                          # leaves the 1.1 deg test cube
    op.config = far
    n_far = len(op.constants.RANGE_RADAR)
    big = torch.zeros((2, 4, n_far), dtype=torch.float32, device='cuda')
    op.simulate_rays(az, np.full(4, 2.0), device_outputs={'ZH': big[0].data_ptr(), 'ZDR': big[1].data_ptr()})
    op.config = over                                        # two clean sweeps behind the bad one
    op.simulate_rays(az, np.full(4, 2.0), device_outputs=ptrs)
    op.simulate_rays(az, np.full(4, 3.0), device_outputs=ptrs)
    with pytest.raises(IndexError):
        op.wait()
    op.wait()                                               # reported once, then cleared
    op._ctx.counters()
    # out-of-domain gates leave no stale coordinates in caller buffers
    op.config = far
    with pytest.raises(IndexError):
        res = op.simulate_rays(az, np.full(4, 2.0))
    op.close()


def test_pinned_outputs_equal_blocking_outputs():
    """outputs_on_device = 2: non-blocking device-to-host copies into the lane's page-locked slab."""
    op, _, _, _ = _op('c3_melt_ice', lanes=2)
    az = np.arange(0., 360., 30.)
    el = np.full(len(az), 4.0)
    ref = op.simulate_rays(az, el)
    for lane in (0, 1):
        out = op.simulate_rays(az, el, pinned=True, lane=lane)
        op.wait(lane)
        for k in FIELDS:
            assert np.array_equal(out[k], ref[k], equal_nan=True), (lane, k)
    # a second pinned sweep on the same lane reuses the slab; geometry comes from the cache
    out = op.simulate_rays(az, np.full(len(az), 7.0), pinned=True, lane=1)
    op.wait(1)
    ref7 = op.simulate_rays(az, np.full(len(az), 7.0))
    for k in FIELDS:
        assert np.array_equal(out[k], ref7[k], equal_nan=True), k
    assert not np.array_equal(ref7['heights'], ref['heights'])
    op.close()


def test_pinned_mask_from_its_one_byte_form_with_sub_beams():
    """With `RadarOperator.compact_mask = True` a pinned result carries the radial mask as cpol_outputs.mask_sum8 (the SUM of the
    sub-beams' codes, one byte per gate) and makes the reference's float64 mask from it on first read (doppler_scatter.py:472-477):
    with 15 and 49 sub-beams -- fractional masks where the sub-beams disagree (rays that leave the model top / start below the
    topography) -- it equals the float64 array the device writes for a blocking call and for a pinned call by default."""
    for name, els in (('c4_subbeams', (2.0, 25.0, 60.0)), ('c4_7x7', (3.0, 40.0))):
        op, _, _, _ = _op(name, lanes=1)
        az = np.arange(0., 360., 45.)
        n_frac = 0
        for e in els:
            el = np.full(len(az), e)
            ref = op.simulate_rays(az, el)
            assert 'mask_sum8' not in ref and ref['mask'].dtype == np.float64
            op.compact_mask = True
            out = op.simulate_rays(az, el, pinned=True)
            op.wait(0)
            op.compact_mask = False
            assert out['mask_sum8'].dtype == np.int8 and out.pending('mask')
            m = out['mask']
            assert m.dtype == np.float64 and np.array_equal(m, ref['mask']) and not out.pending('mask')
            n_frac += int(np.sum((ref['mask'] != np.round(ref['mask']))))
            for k in FIELDS:
                assert np.array_equal(out[k], ref[k], equal_nan=True), (name, e, k)
            plain = op.simulate_rays(az, el, pinned=True)               # (the default: the float64 array from the device)
            op.wait(0)
            assert 'mask_sum8' not in plain and np.array_equal(plain['mask'], ref['mask'])
        assert n_frac > 0, 'no gate where the sub-beams disagree: the fractional masks were not exercised'
        op.close()


def test_too_many_lut_slices_is_refused_by_stage_hydro():
    """cpol_stage_hydro propagates the bucket-scan limit (n_e x n_t summed over the slots)
    instead of leaving a context that cpol_run_sweep would mis-sort."""
    from cosmo_pol_amd import _native as N, hydrometeors as hyd
    op, over, _, luts = _op('c2_rsg')
    conf = op.config
    vi = {v: i for i, v in enumerate(hyd.variable_list(conf))}
    d, table, pre, dnu, aux = hyd.build_hydro('R', '1mom', luts['R'], vi)
    n_e, n_t = int(d.n_e), int(d.n_t)
    big_e = (32768 // n_t) + 8                      # n_e * n_t alone exceeds 1024 * 32 slices
    d2 = copy.copy(d)
    d2.n_e = big_e
    tbl = np.zeros((big_e, n_t, int(d.n_d), 12))
    ctx = N.Context(0)
    with pytest.raises(ValueError):
        ctx.stage_hydro(0, d2, tbl, pre, dnu, aux)
    ctx.close()
    op.close()


def test_spectrum_sensitivity_cut_through_get_ppi():
    """Doppler scheme 3 with the sensitivity cut ON through get_PPI: the spectrum is censored
    bin by bin (10 log10(S) < threshold(r)), not with the gate mask (doppler_scatter.py:839-850)."""
    name = 'd3_rsg'
    op, over, ocube, luts = _op(name, {'radar': {'sensitivity': [20., 10000]}})
    conf = ocfg.make_config(over)
    azs = np.array([200., 20.])
    scan = op.get_PPI(elevations=[4.0], azimuths=azs)
    raw = scan.raw[0]['fields']
    olut = {h: _cases.as_oracle_lut(l) for h, l in luts.items()}
    n_bins_cut = n_gate_cut = n_survivors_at_cut_gates = 0
    for r, az in enumerate(azs):
        subs = beam.interpolate_radial(ocube, conf, az, 4.0)
        o = scatter.radar_observables(subs, olut, conf)
        before = np.isnan(o.values['DSPECTRUM']).sum()
        zh_before = np.isfinite(o.values['ZH'])
        scatter.cut_at_sensitivity([[o]], conf)
        osp = o.values['DSPECTRUM']
        assert np.array_equal(np.isnan(raw['DSPECTRUM'][r]), np.isnan(osp))
        ok = ~np.isnan(osp)
        np.testing.assert_allclose(raw['DSPECTRUM'][r][ok], osp[ok], rtol=2e-5, atol=1e-6 * np.nanmax(osp))
        assert np.array_equal(np.isnan(raw['ZH'][r]), np.isnan(o.values['ZH']))
        assert np.array_equal(np.isnan(raw['RVEL'][r]), np.isnan(o.values['RVEL']))
        n_bins_cut += int(np.isnan(osp).sum() - before)
        cut_gates = zh_before & np.isnan(o.values['ZH'])
        n_gate_cut += int(cut_gates.sum())
        n_survivors_at_cut_gates += int(np.isfinite(osp[cut_gates]).sum())
    assert n_bins_cut > 0 and n_gate_cut > 0
    op.close()


def test_radial_records_of_a_device_result_and_reference_style_cut():
    """to_radials(simulate_rays(..., apply_sensitivity=False)) censored by the oracle's
    cut_at_sensitivity (the reference's list-of-lists call, radar_operator.py:445) equals
    the sweep censored on the device -- i.e. reference-side code that consumes lists of
    Radial runs unchanged on the batched result (INTEGRATION.md level B)."""
    from cosmo_pol_amd.radial import to_radials
    op, over, _, _ = _op('c4_subbeams', {'radar': {'sensitivity': [30., 10000]}})
    conf = ocfg.make_config(over)
    az = np.arange(0., 360., 40.)
    el = np.full(len(az), 4.0)
    plain = op.simulate_rays(az, el, apply_sensitivity=False)
    rads = to_radials(plain, azimuths=az, elevations=el)
    assert len(rads) == len(az) and rads[0].values['ZH'].shape == (plain['ZH'].shape[1],)
    assert np.array_equal(rads[3].dist_profile, plain['dist'][3])
    scatter.cut_at_sensitivity([rads], conf)                # edits `plain` through the row views
    dev = op.simulate_rays(az, el, apply_sensitivity=True)
    n_cut = 0
    for k in ['ZH', 'ZV', 'ZDR', 'KDP', 'DELTA_HV', 'PHIDP', 'RHOHV', 'ATT_H', 'ATT_V', 'RVEL']:
        assert np.array_equal(plain[k], dev[k], equal_nan=True), k
        n_cut += int(np.isnan(dev[k]).sum())
    assert np.isnan(dev['ZH']).sum() > np.isnan(dev['ATT_H']).sum()     # ATT_* are not censored
    scan = op.get_PPI(elevations=[4.0], azimuths=az)
    sweeps = scan.to_radials()
    assert len(sweeps) == 1 and len(sweeps[0]) == len(az)
    assert np.array_equal(sweeps[0][2].values['ZH'], dev['ZH'][2], equal_nan=True)
    op.close()


def test_ice_units_outside_the_lambda_tables_are_summed_directly():
    """1-moment ice: k_psd_ice2 takes the units whose lambdas lie inside the tabulated range, the
    64-item kernel sums the normalisation integrals of the others bin by bin.  CPOL_ICE_FORCE_SUM=1
    (read once per process) sends EVERY unit down that fallback: the reference-pinned radials with
    ice crystals must still hold (one child pytest process)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, CPOL_ICE_FORCE_SUM='1')
    cmd = [sys.executable, '-m', 'pytest', os.path.join(root, 'tests', 'test_gpu_parity.py'), '-m', 'gpu', '-q',
           '-x', '-k', 'c3_melt_ice or d3_1mom_ice_sub or c4_7x7 or q_ml_dop2']
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=root)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-2000:]
    assert '4 passed' in out.stdout


def test_integral_tables_take_the_items_and_the_integrating_kernels_stay_pinned():
    """Default: every item is finished from the integral tables (k_psd_lookup;
    cpol_counters_t.n_table_items) -- 1-D blocks for the gamma-family species and 1-moment ice,
    2-D blocks (wet fraction x rain-partner slope) for the melting species, whose fit is verified
    at build time against the integrating kernel at one off-node point of every block.
    CPOL_ITAB_MELT=0 keeps the melting species on the integrating kernel, CPOL_SUBSUM=0 the
    sub-beam accumulation inside k_final, CPOL_ITAB=0 integrates
    every item over its 1024 diameter bins -- the kernels that also evaluate the table nodes at
    staging time: the reference-pinned parity module must hold in those modes too (child pytest
    processes: the switches are read when the tables are built)."""
    import os
    import subprocess
    import sys
    op, _, _, _ = _op('c3_melt_ice')
    az = np.arange(0., 360., 30.)
    op.simulate_rays(az, np.full(len(az), 4.0))
    c = op._ctx.counters()
    assert c.n_table_items == c.n_valid_items > 0 and c.n_work_units == 0
    chk = op._ctx.debug_read('itab_check', (2, 8), np.float64)[0]
    assert (chk >= 0).all() and ((chk > 0) & (chk < 1e-10)).sum() == 6, chk     # every table passed its gate
    op.close()
    op, _, _, _ = _op('c2_rsg')
    op.simulate_rays(az, np.full(len(az), 4.0))
    c = op._ctx.counters()
    assert c.n_table_items == c.n_valid_items > 0 and c.n_work_units == 0
    op.close()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    base = [sys.executable, '-m', 'pytest', os.path.join(root, 'tests', 'test_gpu_parity.py'), '-m', 'gpu', '-q', '-x']
    out = subprocess.run(base, env=dict(os.environ, CPOL_ITAB='0'), capture_output=True, text=True,
                         timeout=1100, cwd=root)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-2000:]
    assert ' passed' in out.stdout and 'failed' not in out.stdout
    out = subprocess.run(base + ['-k', 'c3_melt_ice or c4_7x7 or q_ml_dop2'], env=dict(os.environ, CPOL_ITAB_MELT='0'),
                         capture_output=True, text=True, timeout=900, cwd=root)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-2000:]
    assert '3 passed' in out.stdout, out.stdout[-500:]
    # ... and the sub-beam accumulation inside k_final, fed by stored columns (CPOL_SUBSUM=0)
    out = subprocess.run(base + ['-k', 'c4_7x7 or c4_subbeams or q_ml_dop2 or c5_2mom_dop2_sub'],
                         env=dict(os.environ, CPOL_SUBSUM='0'), capture_output=True, text=True, timeout=900, cwd=root)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-2000:]
    assert '4 passed' in out.stdout, out.stdout[-500:]
    # ... and k_subbeam_sum with the coefficient rows through the scalar cache (CPOL_SUBSUM_COOP=1: by default
    # only launches far larger than a test radial take that form)
    out = subprocess.run(base + ['-k', 'c4_7x7 or c4_subbeams or q_ml_dop2 or c5_2mom_dop2_sub'],
                         env=dict(os.environ, CPOL_SUBSUM_COOP='1'), capture_output=True, text=True, timeout=900, cwd=root)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-2000:]
    assert '4 passed' in out.stdout, out.stdout[-500:]


def test_table_sets_stay_resident_across_configuration_switches():
    """cpol_hydro_desc.table_id: the operator names every slot of a cached table set; switching
    the radar frequency and back (what get_GPM_swath does twice per call) finds the scattering
    tables and their integral tables still on the device -- no new cache entry, identical results."""
    op, _, _, _ = _op('c2_rsg')
    az = np.arange(0., 360., 45.)
    el = np.full(len(az), 4.0)
    r1 = {k: np.array(v, copy=True) for k, v in op.simulate_rays(az, el).items() if isinstance(v, np.ndarray)}
    c1 = op._ctx.debug_read('cache', (3,), np.float64)
    assert c1[0] == 3 and c1[1] == 3                      # R, S, G: one entry each
    conf = op.config
    conf['radar']['frequency'] = 13.6
    op.config = conf
    r2 = op.simulate_rays(az, el)
    c2 = op._ctx.debug_read('cache', (3,), np.float64)
    assert c2[0] == 6 and c2[1] == 6
    assert np.isfinite(r2['ZH']).any()
    conf = op.config
    conf['radar']['frequency'] = 5.6
    op.config = conf
    r3 = op.simulate_rays(az, el)
    c3 = op._ctx.debug_read('cache', (3,), np.float64)
    assert c3[0] == 6 and c3[1] == 6                      # nothing new was built or uploaded
    for k in ('ZH', 'ZDR', 'KDP', 'RHOHV', 'PHIDP', 'ATT_H', 'RVEL'):
        assert np.array_equal(r3[k], r1[k], equal_nan=True), k
    op.close()


def test_operator_first_then_torch_cuda():
    """A RadarOperator built BEFORE torch is imported, then torch.cuda used in the same process
    (round 2: RuntimeError "No HIP GPUs are available", two HIP runtimes mapped).  Own interpreter:
    the import order is what is tested."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ('import sys\n'
            'for p in (%r, %r + "/oracle", %r + "/tests"): sys.path.insert(0, p)\n'
            'import numpy as np, _cases\n'
            'from cosmo_pol_amd import RadarOperator, _native as N\n'
            'assert "torch" not in sys.modules\n'
            'over = _cases.gen_golden.radial_case_inputs("c2_rsg")[0]\n'
            '_, az, el, _, luts, cube = _cases.radial_case("c2_rsg")\n'
            'op = RadarOperator(config=over, luts=luts, output_variables="only_radar")\n'
            'op.load_model_arrays(cube["data"], cube["zlevels"], cube["proj_info"], cube["resolution"])\n'
            'a = op.simulate_rays([az], [el])\n'
            'import torch\n'
            'z = torch.zeros(8, device="cuda")\n'
            'slab = torch.zeros((1, a["ZH"].shape[1]), dtype=torch.float32, device="cuda")\n'
            'op.simulate_rays([az], [el], device_outputs={"ZH": slab.data_ptr()})\n'
            'op.wait()\n'
            'assert np.array_equal(slab.cpu().numpy(), a["ZH"], equal_nan=True)\n'
            'assert len(N.hip_runtimes_mapped()) == 1, N.hip_runtimes_mapped()\n'
            'op.close()\n'
            'print("ok", float(z.sum()))\n' % (root, root, root))
    r = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and 'ok 0.0' in r.stdout, (r.stdout[-2000:], r.stderr[-2000:])


def test_results_live_in_pooled_pinned_blocks_of_their_own():
    """Host results are views of context-free pinned blocks from the operator's pool: a steady stream
    of equal sweeps allocates nothing new, dropping a result recycles its block, results survive lane
    drops (set_lut), re-staging and close(), and non-blocking sweeps on alternating lanes whose results
    are dropped before anybody waited do not corrupt later ones."""
    import gc
    op, over, _, _ = _op('c3_melt_ice', lanes=2)
    az = np.arange(0., 360., 30.)
    ref4 = op.simulate_rays(az, np.full(len(az), 4.0))
    ref7 = op.simulate_rays(az, np.full(len(az), 7.0))
    keep4 = {k: np.array(v) for k, v in ref4.items() if isinstance(v, np.ndarray)}
    n0 = op._pool.n_alloc
    for i in range(12):
        r = op.simulate_rays(az, np.full(len(az), 4.0 if i % 2 == 0 else 7.0))
        del r
    gc.collect()
    assert op._pool.n_alloc <= n0 + 1, (n0, op._pool.n_alloc)          # dropped blocks were re-used
    # non-blocking sweeps, results dropped while their copies may still be in flight
    for i in range(10):
        op.simulate_rays(az, np.full(len(az), 7.0), pinned=True, lane=i % 2)
    a = op.simulate_rays(az, np.full(len(az), 4.0), pinned=True, lane=0)
    b = op.simulate_rays(az, np.full(len(az), 7.0), pinned=True, lane=1)
    op.wait(0)
    op.wait(1)
    for k in FIELDS:
        assert np.array_equal(a[k], ref4[k], equal_nan=True), k
        assert np.array_equal(b[k], ref7[k], equal_nan=True), k
    assert op._pool.n_alloc <= n0 + 4
    # lanes dropped, tables re-staged, operator closed: the arrays handed out before stay what they were
    op.set_lut()
    assert op._lane_ctx == []
    for k in FIELDS:
        assert np.array_equal(ref4[k], keep4[k], equal_nan=True), k
    op.close()
    gc.collect()
    for k in FIELDS:
        assert np.array_equal(ref4[k], keep4[k], equal_nan=True), k
        assert np.array_equal(a[k], keep4[k], equal_nan=True), k
    ref4['ZH'][0, 0] = 1.0                                              # still writable memory of its own
    del ref4, ref7, a, b
    gc.collect()


def test_placement_helpers_on_the_box():
    """The bus id the library reports names a PCI device of this host; blocks taken next to GPU 0 are
    ordinary page-locked memory (cpol_host_alloc_near) and the pool uses them."""
    import ctypes as C
    from cosmo_pol_amd import _native as N
    info = N.device_numa_info(0)
    assert os.path.isdir(os.path.join('/sys/bus/pci/devices', info['pci']))
    lib = N.load_library()
    h = C.c_void_p()
    assert lib.cpol_host_alloc_near(0, 1 << 20, C.byref(h)) == 0 and h.value
    buf = (C.c_uint8 * (1 << 20)).from_address(h.value)
    buf[0], buf[-1] = 7, 9
    assert (buf[0], buf[-1]) == (7, 9)
    assert lib.cpol_host_free(None, h) == 0
    assert lib.cpol_host_alloc_near(-1, 1 << 20, C.byref(h)) != 0
    assert lib.cpol_host_alloc_near(999, 1 << 20, C.byref(h)) != 0
    pool = N.PinnedPool(0)
    arr, _ = pool.take(3 << 20)
    arr[:] = 1
    assert int(arr.sum()) == arr.size
    del arr
    pool.close()


def test_ray_table_sets_of_the_last_scan_geometries_stay_on_the_device():
    """The library keeps the per-ray tables of the last 8 tagged scan geometries of a context on the device
    (cpol_ray_tables_t.version): 11 elevations in turn, three times round -- more geometries than sets, so sets
    are evicted and uploaded again; then the first few once more (found resident).  Every result must equal the
    result of an operator that uploads its tables on every call (version 0), bit for bit, on both lanes."""
    op, _, _, _ = _op('c4_subbeams', lanes=2)
    ref, _, _, _ = _op('c4_subbeams')
    ref.reuse_device_tables = False
    az = np.arange(0.0, 360.0, 30.0)
    elevations = [1.0 + 0.7 * k for k in range(11)]
    want = {}
    for e in elevations:
        want[e] = ref.simulate_rays(az, np.full(len(az), e))
    k = 0
    for e in elevations * 3 + elevations[:4] + elevations[:4]:
        got = op.simulate_rays(az, np.full(len(az), e), lane=k % 2)
        k += 1
        for f in ('ZH', 'ZDR', 'KDP', 'PHIDP', 'RVEL', 'mask', 'lats', 'heights'):
            assert np.array_equal(got[f], want[e][f], equal_nan=True), (f, e, k)
    assert np.isfinite(want[elevations[2]]['ZH']).sum() > 50
    op.close()
    ref.close()
