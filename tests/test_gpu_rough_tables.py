"""The integral tables on NON-SMOOTH scattering tables.

The default PSD stage evaluates degree-10 polynomials of the PSD-integrated scattering entries
(k_psd_lookup / k_subbeam_sum) instead of integrating 1024 diameter bins.  Until round 3 that was
only ever exercised on tables that are smooth in D.  Here: white noise with sign changes,
resonance ripples and spikes, and columns whose integral nearly cancels (tests/_rough.py), for
R / S / G / I and mS / mG, 1- and 2-moment, at C, Ku and Ka band:

    HIP with the tables (default)   vs   HIP with CPOL_ITAB=0 (every item integrated bin by bin)
                                    vs   the CPU oracle (reference algorithm: gather + einsum)

plus the build-time accuracy gate (k_itab_check1 / k_itab_check2): every table either passes at
1e-10 on the block's scale or the species demonstrably falls back to the integrating kernels.

Tolerance: PURE 1e-5 relative on every PSD-integrated entry sz_integ[gate, hydrometeor, column]
(what the tables produce); the observables derived from them at pure 1e-5 wherever the float32
arithmetic of get_pol_from_sz itself is well conditioned (sign-changing tables make ZH a
difference of float32 numbers: there a 1-ulp change of an operand is amplified by the
cancellation factor, in the reference as well)."""
import json
import os

import numpy as np
import pytest

import _cases
import _rough
from cosmo_pol_oracle import beam, scatter
from cosmo_pol_oracle import config as ocfg

pytestmark = pytest.mark.gpu
RTOL = 1e-5
CASES = ['c3_melt_ice', 'c4_subbeams', 'c5_2mom', 'c5_ka_2mom']


def _run_hip(over, luts, cube, az, el, itab):
    from cosmo_pol_amd import RadarOperator
    old = os.environ.pop('CPOL_ITAB', None)
    if not itab:
        os.environ['CPOL_ITAB'] = '0'            # read when the tables are (re)built
    try:
        op = RadarOperator(config=over, luts=luts, output_variables='only_radar')
        op.load_model_arrays(cube['data'], cube['zlevels'], cube['proj_info'], cube['resolution'])
        op._ctx.enable_debug(True)
        res = op.simulate_rays([az], [el], apply_sensitivity=False)
        n_gates = res['ZH'].shape[1]
        n_hyd = len(op._staged_hydro)
        out = {k: np.array(v[0]) for k, v in res.items() if isinstance(v, np.ndarray)}
        out['sz_integ'] = op._ctx.debug_read('sz_integ', (n_gates, n_hyd, 12), np.float32)
        out['sz_total'] = op._ctx.debug_read('sz_total', (n_gates, 12), np.float32)
        cnt = op._ctx.counters()
        out['n_table_items'], out['n_valid_items'] = int(cnt.n_table_items), int(cnt.n_valid_items)
        out['report'] = op._ctx.itab_report()
        out['detail'] = [op._ctx.itab_detail(j) for j in range(n_hyd)]
        out['hydro'] = list(op._staged_hydro)
        op.close()
        return out
    finally:
        os.environ.pop('CPOL_ITAB', None)
        if old is not None:
            os.environ['CPOL_ITAB'] = old


def _record(rec):
    print('ROUGH', json.dumps(rec))
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'gpurun_out')
    try:
        os.makedirs(out, exist_ok=True)
        with open(os.path.join(out, 'rough_table_records.jsonl'), 'a') as f:
            f.write(json.dumps(rec) + '\n')
    except OSError:
        pass


def _worst_rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    ok = np.isfinite(a) & np.isfinite(b) & (b != 0)
    return float(np.max(np.abs(a[ok] - b[ok]) / np.abs(b[ok]))) if ok.any() else 0.0


def _amplification(szi, szt):
    """Per observable and gate: how much a relative change of the float32-stored PSD integrals
    sz_integ[gate, hydrometeor, column] can be amplified by the float32 arithmetic behind them
    (sum over hydrometeors, then ZH ~ s0 - s1 - s2 + s3, ZV ~ s0 + s1 + s2 + s3, attenuations ~
    s11 / s9): sum of |operands| over |result|.  1 = no cancellation."""
    a = np.abs(np.nan_to_num(szi.astype(np.float64))).sum(axis=1)          # [gate, column]
    t = np.nan_to_num(szt.astype(np.float64))
    with np.errstate(divide='ignore', invalid='ignore'):
        amp = {'ZH': a[:, :4].sum(axis=1) / np.abs(t[:, 0] - t[:, 1] - t[:, 2] + t[:, 3]),
               'ZV': a[:, :4].sum(axis=1) / np.abs(t[:, 0] + t[:, 1] + t[:, 2] + t[:, 3]),
               'ATT_H': a[:, 11] / np.abs(t[:, 11]), 'ATT_V': a[:, 9] / np.abs(t[:, 9])}
    amp['ZDR'] = np.maximum(amp['ZH'], amp['ZV'])
    return {k: np.nan_to_num(v, nan=np.inf) for k, v in amp.items()}, a


@pytest.mark.parametrize('kind', _rough.KINDS)
@pytest.mark.parametrize('name', CASES)
def test_rough_tables_itab_vs_direct_vs_oracle(name, kind):
    conf, az, el, ocube, luts, cube = _cases.radial_case(name)
    over = _cases.gen_golden.radial_case_inputs(name)[0]
    rl = _rough.roughen_all(luts, kind, frequency=conf['radar']['frequency'], scheme=conf['microphysics']['scheme'])
    on = _run_hip(over, rl, cube, az, el, itab=True)
    off = _run_hip(over, rl, cube, az, el, itab=False)
    subs = beam.interpolate_radial(ocube, conf, az, el)
    o = scatter.radar_observables(subs, {h: _cases.as_oracle_lut(l) for h, l in rl.items()}, conf,
                                  return_sz=True)
    hl = on['hydro']
    assert hl == ocfg.hydrometeor_list(conf)
    rep = on['report']
    # ---- the accuracy gate: every slot has a verdict; accepted tables passed at 1e-10 ----
    accepted, ranges = [], {}
    for j, h in enumerate(hl):
        chk = float(rep['check'][j])
        assert chk != 0.0, 'slot %s has no verdict' % h
        if chk > 0:
            assert chk < 1e-10, (h, chk)
            d = on['detail'][j]
            if d is not None:                    # 1-D table: the run of lambda panels that passed the gate
                lo, hi = d['accepted_panels']
                assert np.all(d['by_pan'][lo:hi] < 1e-10) and 2 * (hi - lo) >= d['n_pan'], (h, lo, hi)
                # all but the last panels -- except for 'alternating', where node values that are
                # rounding noise of a 1e10-fold cancellation must make the gate cut the range
                # (and for 'mie': a Mie table is accepted where it passes, the rest of its lambda range falls back)
                # ('random_sign': since the second check point exists -- u = 0.96, round 4 -- the white-noise rain
                # table loses its first panels too: 2.3e-10 in panel 8, which the mid-panel point had passed)
                assert kind in ('alternating', 'mie', 'random_sign') or (lo == 0 and hi >= d['n_pan'] - 2), (h, lo, hi)
                ranges[h] = (lo, hi, d['n_pan'])
            accepted.append(h)
    assert off['n_table_items'] == 0 and off['n_valid_items'] == on['n_valid_items']
    assert np.all(off['report']['check'] == 0)
    if accepted and kind != 'alternating':       # ('alternating': the items may all lie in panels the gate cut)
        assert on['n_table_items'] > 0
    # ---- PSD-integrated entries: pure 1e-5, tables on and off, against the oracle ----
    # ('alternating': the odd columns are 1024-term float64 sums that cancel up to 1e10-fold, i.e. defined
    # to ~1e-6 .. 1e-4 of their value whatever the summation order -- NumPy's pairwise einsum and the
    # device's chunked sums differ by that much; the point of that kind is the gate, checked above)
    RTOL = 1e-3 if kind == 'alternating' else 1e-5              # (shadows the module's 1e-5 inside this test)
    assert np.isfinite(o.sz_integ).sum() > 200, 'the case was not exercised'
    _cases.assert_close_nan(on['sz_integ'], o.sz_integ, rtol=RTOL, name='tables on: sz_integ')
    _cases.assert_close_nan(off['sz_integ'], o.sz_integ, rtol=RTOL, name='tables off: sz_integ')
    # ---- sum over hydrometeors (float32): 1e-5 of the operands ----
    opscale = np.nan_to_num(np.abs(o.sz_integ.astype(np.float64))).sum(axis=1)
    _cases.assert_close_nan(on['sz_total'], o.sz_total, rtol=RTOL, atol=RTOL * opscale, name='tables on: sz_total')
    # ---- observables: pure 1e-5 where the float32 arithmetic behind them is well conditioned;
    # K_DP (always a difference of near-equal sums) to 1e-5 of its operands ----
    amp, opabs = _amplification(o.sz_integ, o.sz_total)
    finite = np.isfinite(o.values['ZH'])
    worst, n_cmp = {}, {}
    for k in ['ZH', 'ZV', 'ZDR', 'ATT_H', 'ATT_V']:
        well = amp[k] < 30.0
        for tag, r in (('on', on), ('off', off)):
            assert np.array_equal(np.isnan(r[k]), np.isnan(o.values[k])), (tag, k)
            _cases.assert_close_nan(r[k][well], o.values[k][well], rtol=RTOL, name='%s: %s' % (tag, k))
        worst[k] = _worst_rel(on[k][well], o.values[k][well])
        n_cmp[k] = int((well & finite).sum())
    assert n_cmp['ZH'] >= 0.3 * finite.sum() and n_cmp['ATT_H'] >= 0.3 * finite.sum(), (n_cmp, int(finite.sum()))
    if kind == 'alternating':
        # the gate engaged: part of the items went to the integrating kernels
        assert any(r[0] > 0 or r[1] < r[2] - 2 for r in ranges.values()) or len(accepted) < len(hl), ranges
        assert on['n_table_items'] <= on['n_valid_items']
    from cosmo_pol_oracle import constants as OK
    kdp_atol = RTOL * 1e-3 * (180.0 / np.pi) * OK.Derived(conf).WAVELENGTH * (opabs[:, 8] + opabs[:, 10])
    for tag, r in (('on', on), ('off', off)):
        _cases.assert_close_nan(r['KDP'], o.values['KDP'], rtol=RTOL, atol=kdp_atol, name=tag + ': KDP')
    worst['KDP'] = _worst_rel(on['KDP'], o.values['KDP'])
    if kind == 'resonance':                      # positive tables: the remaining observables as well
        res_km = conf['radar']['radial_resolution'] / 1000.
        well = amp['ZDR'] < 30.0
        for tag, r in (('on', on), ('off', off)):
            _cases.assert_close_nan(r['RHOHV'][well], o.values['RHOHV'][well], rtol=RTOL, name=tag + ': RHOHV')
            _cases.assert_close_nan(r['DELTA_HV'], o.values['DELTA_HV'], rtol=RTOL, atol=RTOL * np.pi,
                                    name=tag + ': DELTA_HV')
            _cases.assert_close_nan(r['PHIDP'], o.values['PHIDP'], rtol=RTOL,
                                    atol=np.cumsum(2 * kdp_atol) * res_km + RTOL * np.pi, name=tag + ': PHIDP')
        worst['RHOHV'] = _worst_rel(on['RHOHV'][well], o.values['RHOHV'][well])
    _record({'case': name, 'kind': kind, 'hydro': hl,
             'itab_check': [float(x) for x in rep['check'][:len(hl)]],
             'itab_check_edge': [float(x) for x in rep['check_edge'][:len(hl)]],
             'accepted': accepted, 'accepted_panels': ranges, 'n_table_items': on['n_table_items'], 'n_valid_items': on['n_valid_items'],
             'sz_integ_worst_rel_on': _worst_rel(on['sz_integ'], o.sz_integ),
             'sz_integ_worst_rel_off': _worst_rel(off['sz_integ'], o.sz_integ),
             'sz_integ_worst_rel_on_vs_off': _worst_rel(on['sz_integ'], off['sz_integ']),
             'gates_compared_pure_1e-5': n_cmp, 'gates_finite': int(finite.sum()), 'observables_worst_rel': worst,
             'build_ms': [float(x) for x in rep['build_ms'][:len(hl)]],
             'check_ms': [float(x) for x in rep['check_ms'][:len(hl)]]})


def test_rejected_table_falls_back_to_the_integrating_kernels(monkeypatch):
    """CPOL_ITAB_MAX_DEV below anything a float64 polynomial can reach: every table is rejected by
    the gate, itab_check turns negative, no item is looked up -- and the results are still right."""
    name = 'c3_melt_ice'
    conf, az, el, ocube, luts, cube = _cases.radial_case(name)
    over = _cases.gen_golden.radial_case_inputs(name)[0]
    monkeypatch.setenv('CPOL_ITAB_MAX_DEV', '1e-40')
    r = _run_hip(over, luts, cube, az, el, itab=True)
    monkeypatch.delenv('CPOL_ITAB_MAX_DEV')
    hl = r['hydro']
    chk = r['report']['check'][:len(hl)]
    assert np.all(chk < 0), chk
    assert r['n_table_items'] == 0 and r['n_valid_items'] > 200
    good = _run_hip(over, luts, cube, az, el, itab=True)
    assert np.all(good['report']['check'][:len(hl)] > 0) and np.all(good['report']['check'][:len(hl)] < 1e-10)
    assert good['n_table_items'] == good['n_valid_items'] == r['n_valid_items']
    subs = beam.interpolate_radial(ocube, conf, az, el)
    o = scatter.radar_observables(subs, {h: _cases.as_oracle_lut(l) for h, l in luts.items()}, conf,
                                  return_sz=True)
    _cases.assert_close_nan(r['sz_integ'], o.sz_integ, rtol=RTOL, name='fallback: sz_integ')
    _cases.assert_close_nan(good['sz_integ'], o.sz_integ, rtol=RTOL, name='tables: sz_integ')


def test_check_costs_little_on_full_size_tables():
    """Full-size R, S, G tables (46 x 27|39 slices): the accuracy gate (the two check items of every block of
    11 nodes, compared inside k_itab_fit) takes < 8 % of cpol_prepare, and the smooth bench tables pass far
    below 1e-10 at both check points (mid-panel and between the last two nodes)."""
    import time
    import bench
    from cosmo_pol_amd import RadarOperator, synthetic
    luts = synthetic.make_all_luts(('R', 'S', 'G'), 5.6, '1mom')
    t0 = time.perf_counter()
    op = RadarOperator(config=bench.bench_config(False), luts=luts, output_variables='only_radar')
    op._ctx.synchronize()
    t_prepare = 1e3 * (time.perf_counter() - t0)             # stage_hydro + cpol_prepare of the three slots
    rep = op._ctx.itab_report()
    det = [op._ctx.itab_detail(j) for j in range(3)]
    op.close()
    for d in det:
        # the gate drops at most the last panels (exp(-lambda D^nu) of the bins behind the first goes
        # subnormal there: lambda D_0^nu > 650, a mass density far below anything a model cell holds)
        lo, hi = d['accepted_panels']
        assert lo == 0 and hi >= d['n_pan'] - 2, d['accepted_panels']
        assert np.all(d['by_pan'][lo:hi] < 1e-10)
    chk, ms_chk, ms_all = rep['check'][:3], rep['check_ms'][:3], rep['build_ms'][:3]
    _record({'case': 'bench R,S,G full-size tables', 'itab_check': [float(x) for x in chk],
             'itab_check_edge': [float(x) for x in rep['check_edge'][:3]],
             'build_ms': [float(x) for x in ms_all], 'check_ms': [float(x) for x in ms_chk],
             'accepted_panels': [d['accepted_panels'] + (d['n_pan'],) for d in det],
             'worst_by_function': [[float(x) for x in d['by_fn']] for d in det],
             'set_lut_wall_ms': t_prepare})
    assert np.all(chk > 0) and np.all(chk < 1e-10), chk
    edge = rep['check_edge'][:3]
    assert np.all(edge > 0) and np.all(edge <= chk), (edge, chk)        # (`check` = the maximum over both points)
    # the gate = two more items per block of 11 nodes (the comparison itself rides in the fit kernel)
    assert ms_chk.sum() < 0.08 * t_prepare, (ms_chk, t_prepare)
    assert ms_chk.sum() < 0.17 * ms_all.sum(), (ms_chk, ms_all)
