"""The path bench.py's lines TIME, pinned directly (round-5 review, "next" item 1).

bench.py's c2 step is not the default `simulate_rays` call of the other tests: a context with three lanes (sweeps in
flight side by side), device outputs, the one sub-beam of every sweep on the coordinate polynomials of its resident table
set, `k_gate1_ray` + `k_scan_rays` instead of `k_gate1` + `k_final`.  Each piece is compared with the other launch forms
elsewhere (tests/test_gpu_edges.py, test_gpu_fullsize.py); here THAT combination runs on the full configs[1] cube exactly
as `bench.run_c2::step_hbm` submits it (operator and inputs from `bench.make_inputs`), the launch forms it took are read
back, and what it left in HBM is compared

* with the oracle (the reference's algorithm, cosmo_pol/scatter/doppler_scatter.py:400-416 and everything under it) on nine
  sampled rays of every elevation at the PURE relative 1e-5 north_star states -- NaN patterns equal; the three
  phase-like variables (KDP, PHIDP, DELTA_HV: differences of float32-stored sums that cross zero) are allowed their
  operand scale only gate by gate, every such gate counted and recorded (profiles/r6_parity_records.jsonl);
* bit for bit with the host-output path of the same elevations (the reference's hand-over).

The same for the c3 step (`bench.run_c3::volume`: five lanes, page-locked host outputs of a repeating volume scan).
"""
import numpy as np
import pytest

import _cases
import bench
from cosmo_pol_oracle import beam, scatter
from cosmo_pol_oracle import config as ocfg

pytestmark = pytest.mark.gpu
RTOL = 1e-5
PHASE_LIKE = ('KDP', 'PHIDP', 'DELTA_HV')
SAMPLE_RAYS = (0, 37, 90, 123, 181, 222, 275, 301, 359)


def _operator(workload):
    from cosmo_pol_amd import RadarOperator
    conf, hyds, cube, luts = bench.make_inputs(workload, False)
    op = RadarOperator(config=conf, luts=luts, output_variables='only_radar')      # (as bench.main: default lanes argument)
    op.load_model_arrays(cube['data'], cube['zlevels'], cube['proj_info'], cube['resolution'])
    oconf = ocfg.make_config(conf)
    oc = beam.ModelCube({n: cube['data'][n] for n in _cases.ORDER if n in cube['data']}, cube['zlevels'],
                        cube['proj_info'], cube['resolution'], [n for n in _cases.ORDER if n in cube['data']])
    ol = {h: _cases.as_oracle_lut(l) for h, l in luts.items()}
    return op, oconf, oc, ol


def _compare_ray(case, got, o, szt, oconf, stats):
    """One ray's nine fields against the oracle's: pure 1e-5; a phase-like gate beyond it must sit inside its operand scale
    (test_gpu_parity._pol_tolerances) and is counted."""
    from test_gpu_parity import _pol_tolerances
    for k in bench.RADAR_FIELDS:
        a, b = np.asarray(got[k], dtype=np.float64), np.asarray(o.values[k], dtype=np.float64)
        assert np.array_equal(np.isnan(a), np.isnan(b)), '%s %s: NaN pattern' % (case, k)
        ok = np.isfinite(b)
        err = np.abs(a[ok] - b[ok])
        beyond = err > RTOL * np.abs(b[ok])
        st = stats.setdefault(k, {'n': 0, 'beyond_pure': 0, 'worst_rel': 0.0})
        st['n'] += int(ok.sum())
        nz = b[ok] != 0
        if nz.any():
            st['worst_rel'] = max(st['worst_rel'], float(np.max(err[nz] / np.abs(b[ok][nz]))))
        if beyond.any():
            assert k in PHASE_LIKE, '%s %s: %d gates beyond the pure 1e-5 (worst %.3g)' % (case, k, beyond.sum(), st['worst_rel'])
            atol = np.broadcast_to(_pol_tolerances(k, o, szt, oconf), b.shape)[ok]
            assert np.all(err[beyond] <= RTOL * np.abs(b[ok][beyond]) + atol[beyond]), '%s %s: beyond the operand scale' % (case, k)
            st['beyond_pure'] += int(beyond.sum())


def _record(case, stats):
    import json
    import os
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'gpurun_out')
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, 'headline_parity_records.jsonl'), 'a') as f:
        for k, st in stats.items():
            rec = dict(case=case, var=k, n=st['n'], worst_rel=st['worst_rel'], gates_beyond_pure_1e_5=st['beyond_pure'])
            print('HEADLINE-PARITY', json.dumps(rec))
            f.write(json.dumps(rec) + '\n')


def _oracle_ray(oc, oconf, ol, az, el):
    subs = beam.interpolate_radial(oc, oconf, float(az), float(el))
    o = scatter.radar_observables(subs, ol, oconf, return_sz=True)
    szt = np.nan_to_num(o.sz_total.astype(np.float64))
    scatter.cut_at_sensitivity([o], oconf)          # (simulate_rays applies the sensitivity cut by default, as get_PPI does)
    return o, szt


def test_c2_headline_step_vs_oracle_and_host_output_path():
    import torch
    op, oconf, oc, ol = _operator('c2')
    n_lanes, n_cycle = 3, 8
    lanes = [op._lane(i) for i in range(n_lanes)]
    az = np.arange(0, 360, 1.0)
    els = [np.full(len(az), e) for e in bench.C2_ELEVATIONS]
    n_rays, n_gates = len(az), len(op.constants.RANGE_RADAR)
    assert (n_rays, n_gates) == (360, 500)
    # (bench keeps max(2, lanes) slabs and overwrites them; here one per elevation, so that all eight can be read back)
    slabs = [torch.full((len(bench.RADAR_FIELDS), n_rays, n_gates), -7.0, dtype=torch.float32, device='cuda') for _ in range(n_cycle)]
    dev_outs = [{k: sl[i].data_ptr() for i, k in enumerate(bench.RADAR_FIELDS)} for sl in slabs]
    rvel = [torch.full((n_rays, n_gates), -7.0, dtype=torch.float64, device='cuda') for _ in range(n_cycle)]
    for d_o, rv in zip(dev_outs, rvel):
        d_o['RVEL'] = rv.data_ptr()                  # (the tenth field of the headline step)
    forms = []
    for cycle in range(3):                           # (bench: two set-up steps, warm-up, then the timed cycles)
        for k in range(n_cycle):
            op.simulate_rays(az, els[k % n_cycle], device_outputs=dev_outs[k], lane=k % n_lanes)
            forms.append(lanes[k % n_lanes].launch_forms())
    for i in range(n_lanes):
        op.wait(i)
    torch.cuda.synchronize()
    # ---- what ran: the headline's launch forms, on every sweep of the last two cycles ----
    for f in forms[n_cycle:]:
        assert f['gate1_ray'] == 1 and f['gate1'] == 1, f         # k_interp_sweep -> k_gate1_ray -> k_scan_rays
        assert f['poly_central'] == 1, f                          # the coordinate polynomials of the resident table set
        assert f['n_sub'] == 1 and f['lanes_alive'] >= 2, f
    got = [sl.cpu().numpy() for sl in slabs]
    got_rvel = [rv.cpu().numpy() for rv in rvel]
    # ---- against the oracle: nine rays of every elevation, pure 1e-5 ----
    stats, n_valid = {}, 0
    for k in range(n_cycle):
        for r in SAMPLE_RAYS:
            o, szt = _oracle_ray(oc, oconf, ol, az[r], bench.C2_ELEVATIONS[k])
            _compare_ray('c2 el %.2f ray %d' % (bench.C2_ELEVATIONS[k], r),
                         {f: got[k][i, r] for i, f in enumerate(bench.RADAR_FIELDS)}, o, szt, oconf, stats)
            _cases.assert_close_nan(got_rvel[k][r], o.values['RVEL'], rtol=RTOL, atol=2e-4, name='RVEL el %d ray %d' % (k, r))
            n_valid += int(np.isfinite(o.values['ZH']).sum())
    assert n_valid > 10000
    _record('c2_headline_step', stats)
    # ---- bit for bit against the host-output path of the same elevations (all 15 arrays to host memory) ----
    for k in range(n_cycle):
        host = op.simulate_rays(az, els[k])
        for i, f in enumerate(bench.RADAR_FIELDS):
            assert host[f].dtype == np.float32
            assert np.array_equal(host[f], got[k][i], equal_nan=True), (f, k)
        assert np.array_equal(host['RVEL'], got_rvel[k], equal_nan=True), ('RVEL', k)
        assert np.isfinite(host['lats']).all() and host['mask'].shape == (n_rays, n_gates) and host['mask'].dtype == np.float64
    op.close()
    # ---- and against a context WITHOUT lanes: the four-launch sequence (k_gate1_species + k_final: one workgroup barrier, no
    # presence words, no tile rotation, no tickets) on the whole sweep, bit for bit ----
    from cosmo_pol_amd import RadarOperator
    conf, hyds, cube, luts = bench.make_inputs('c2', False)
    op1 = RadarOperator(config=conf, luts=luts, output_variables='only_radar', lanes=1)
    op1.load_model_arrays(cube['data'], cube['zlevels'], cube['proj_info'], cube['resolution'])
    for k in (0, 5):
        one = op1.simulate_rays(az, els[k])
        f1 = op1._ctx.launch_forms()
        assert f1['gate1_ray'] == 0 and f1['gate1'] == 1, f1
        for i, f in enumerate(bench.RADAR_FIELDS):
            assert np.array_equal(one[f], got[k][i], equal_nan=True), (f, k)
        assert np.array_equal(one['RVEL'], got_rvel[k], equal_nan=True), ('RVEL', k)
    op1.close()


def test_c3_step_vs_oracle_and_blocking_path():
    import torch
    op, oconf, oc, ol = _operator('c3')
    az = np.arange(0, 360, 1.0)
    n_el = len(bench.C4_ELEVATIONS)
    els = [np.full(len(az), e) for e in bench.C4_ELEVATIONS]
    for i in range(n_el):
        op._lane(i)
    vol = None
    for _ in range(3):
        vol = [op.simulate_rays(az, els[e], pinned=True, lane=e) for e in range(n_el)]
        for i in range(n_el):
            op.wait(i)
    torch.cuda.synchronize()
    stats, n_valid, n_melt = {}, 0, 0
    for e in range(n_el):
        for r in SAMPLE_RAYS:
            rr = (r + 41 * e) % 360
            o, szt = _oracle_ray(oc, oconf, ol, az[rr], bench.C4_ELEVATIONS[e])
            _compare_ray('c3 el %.1f ray %d' % (bench.C4_ELEVATIONS[e], rr), {f: vol[e][f][rr] for f in bench.RADAR_FIELDS},
                         o, szt, oconf, stats)
            assert np.array_equal(vol[e]['mask'][rr], o.mask)
            n_valid += int(np.isfinite(o.values['ZH']).sum())
        n_melt += int(op._lane(e).counters().n_valid_items)
    assert n_valid > 10000
    _record('c3_step', stats)
    # the blocking one-sweep call on lane 0 (resident tables): same bits
    for e in range(n_el):
        one = op.simulate_rays(az, els[e])
        for f in bench.RADAR_FIELDS + ['RVEL', 'mask', 'lats', 'lons', 'dist', 'heights']:
            assert np.array_equal(one[f], vol[e][f], equal_nan=True), (f, e)
    op.close()
