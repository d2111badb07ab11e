#!/usr/bin/env python
"""Re-runs ONE case of tools/fuzz_parity.py and prints where the Doppler spectrum differs.
   python tools/fuzz_case.py <seed> <case>"""
import copy
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'oracle'), os.path.join(ROOT, 'tests'), os.path.join(ROOT, 'tools')):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402


def main():
    seed, case = int(sys.argv[1]), int(sys.argv[2])
    import fuzz_parity as F
    import _cases
    from cosmo_pol_amd import RadarOperator, synthetic
    from cosmo_pol_oracle import beam, scatter
    from cosmo_pol_oracle import config as ocfg
    rng = np.random.default_rng(seed)
    for _ in range(case + 1):          # same draws, in the same order, as fuzz_parity.main
        over, two = F.draw(rng)
        azs = rng.uniform(0, 360, 2)
        els = rng.uniform(0.5, 30, 2)
        rng.random()                                   # cut
        if rng.random() < 0.3:                         # nyquist
            rng.uniform(0.5, 6.0)
    conf = ocfg.make_config(over)
    hl = ocfg.hydrometeor_list(conf)
    cube = synthetic.small_test_cube(hydrometeors=('R', 'S', 'G', 'I'), two_moment=two, **_cases.gen_golden.CUBE_KW)
    order = _cases.ORDER_2MOM if two else _cases.ORDER
    ocube = beam.ModelCube({n: cube['data'][n].copy() for n in order}, cube['zlevels'], cube['proj_info'],
                           cube['resolution'], order)
    luts = {h: _cases.synthetic_lut(h, conf['radar']['frequency'], conf['microphysics']['scheme']) for h in hl}
    olut = {h: _cases.as_oracle_lut(l) for h, l in luts.items()}
    op = RadarOperator(config=copy.deepcopy(over), luts=luts, output_variables='only_radar', lanes=1)
    op.load_model_arrays(cube['data'], cube['zlevels'], cube['proj_info'], cube['resolution'])
    res = op.simulate_rays(azs, els, apply_sensitivity=False)
    print(over)
    for r in range(2):
        subs = beam.interpolate_radial(ocube, conf, azs[r], els[r])
        o = scatter.radar_observables(subs, olut, conf)
        if 'DSPECTRUM' not in o.values:
            continue
        got, osp = res['DSPECTRUM'][r], o.values['DSPECTRUM']
        bad = np.argwhere(np.abs(got - osp) > 1e-6 * np.nanmax(osp) + 2e-5 * np.abs(osp))
        print('ray', r, 'differing (gate, bin):', bad.tolist())
        for g, v in bad:
            print('  gate %d bin %d gpu %.9g oracle %.9g  row sums %.9g %.9g' % (g, v, got[g, v], osp[g, v],
                                                                              got[g].sum(), osp[g].sum()))
            nz = np.nonzero(osp[g])[0]
            print('  oracle nonzero bins', nz.tolist(), ' gpu nonzero bins', np.nonzero(got[g])[0].tolist())
    op.close()


if __name__ == '__main__':
    main()
