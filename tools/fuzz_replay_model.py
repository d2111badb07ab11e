#!/usr/bin/env python
"""Replays one case of tools/fuzz_parity.py and prints, for the antenna-averaged model variables, where the product and the
oracle differ most: gate, the sub-beams' weights and values there.   python tools/fuzz_replay_model.py <n_cases> <seed> <case>"""
import copy
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'oracle'), os.path.join(ROOT, 'tests'), os.path.join(ROOT, 'tools')):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402


def main():
    import _cases
    import fuzz_parity as F
    from cosmo_pol_amd import RadarOperator, synthetic
    from cosmo_pol_oracle import beam
    from cosmo_pol_oracle import config as ocfg
    n, seed, want = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
    for case, over, two, azs, els, cut, nyq in F.draws(n, seed):
        if case != want:
            continue
        conf = ocfg.make_config(over)
        hl = ocfg.hydrometeor_list(conf)
        cube = synthetic.small_test_cube(hydrometeors=('R', 'S', 'G', 'I'), two_moment=two, **_cases.gen_golden.CUBE_KW)
        order = _cases.ORDER_2MOM if two else _cases.ORDER
        oc = beam.ModelCube({k: cube['data'][k].copy() for k in order}, cube['zlevels'], cube['proj_info'], cube['resolution'], order)
        luts = {h: _cases.synthetic_lut(h, conf['radar']['frequency'], conf['microphysics']['scheme']) for h in hl}
        op = RadarOperator(config=copy.deepcopy(over), luts=luts, output_variables='all', lanes=1)
        op.load_model_arrays(cube['data'], cube['zlevels'], cube['proj_info'], cube['resolution'])
        op._ctx.enable_debug(True)
        res = op.simulate_rays(azs, els, apply_sensitivity=cut)
        n_sub, ngt = res['n_sub'], res['ZH'].shape[1]
        try:
            wg = op._ctx.debug_read('sub_wgate', (2, n_sub, ngt), np.float64)
        except Exception as e:
            wg = None
            print('no sub_wgate:', e)
        for r in range(2):
            subs = beam.interpolate_radial(oc, conf, azs[r], els[r])
            integ = beam.integrate_subbeams(subs)
            ng = len(subs[0].dist_profile)
            W = np.array([np.broadcast_to(sb.quad_weight, (ng,)) for sb in subs])
            for i, nm in enumerate(op._staged_vars):
                a, b = res['model_vars'][i][r], np.asarray(integ.values[nm], dtype=np.float64)
                with np.errstate(invalid='ignore', divide='ignore'):
                    rel = np.nan_to_num(np.abs(a - b) / np.abs(b))
                g = int(np.argmax(rel))
                if rel[g] > 1e-12:
                    V = np.array([np.asarray(sb.values[nm], dtype=np.float64)[g] for sb in subs])
                    M = np.array([sb.mask[g] for sb in subs])
                    print('ray %d %s gate %d: product %.17g oracle %.17g rel %.3e' % (r, nm, g, a[g], b[g], rel[g]))
                    print('   weights', W[:, g].tolist())
                    print('   values ', V.tolist())
                    print('   masks  ', M.tolist(), ' total weight', W[:, g].sum())
                    if wg is not None:
                        print('   device weights - oracle weights', (wg[r][:, g] - W[:, g]).tolist())
                    # per sub-beam at that gate: the device's interpolated value and float32 grid coordinates against the oracle's,
                    # in the default form and with every sub-beam on the long form (CPOL_DEBUG_EXACT_SUBBEAMS)
                    from cosmo_pol_amd import _native as N
                    for flags, tag in ((0, 'default'), (N.DEBUG_EXACT_SUBBEAMS, 'long form')):
                        op.debug_flags = flags
                        op.simulate_rays(azs, els, apply_sensitivity=cut)
                        dv = op._ctx.debug_read('sub_values', (len(op._staged_vars), 2, n_sub, ngt), np.float32)[i, r, :, g]
                        dc = op._ctx.debug_read('sub_coords', (2, n_sub, ngt, 2), np.float32)[r, :, g]
                        oc_rc = np.array([beam.gate_coordinates(oc, conf['radar']['coords'], sb.quad_pt[0], sb.dist_profile)[2][g] for sb in subs], dtype=np.float32)
                        print('   [%s] device value - oracle value (float32 ulps of the value):' % tag,
                              [None if not np.isfinite(x) else float(np.round((np.float64(d) - x) / np.spacing(np.float32(x)), 2)) for d, x in zip(dv, V)])
                        print('   [%s] device coords - oracle coords (ulps): lat' % tag,
                              [float((np.float64(a_) - np.float64(b_)) / np.spacing(np.float32(b_))) for a_, b_ in zip(dc[:, 0], oc_rc[:, 0])],
                              'lon', [float((np.float64(a_) - np.float64(b_)) / np.spacing(np.float32(b_))) for a_, b_ in zip(dc[:, 1], oc_rc[:, 1])])
                    op.debug_flags = 0
        op.close()


if __name__ == '__main__':
    main()
