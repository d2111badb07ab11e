#!/usr/bin/env python
"""Random GPM-DPR swaths (spaceborne geometry, ragged rays, 2-moment microphysics): HIP path
vs the oracle.   python tools/fuzz_gpm.py [n_cases] [seed]"""
import os
import sys
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'oracle'), os.path.join(ROOT, 'tests')):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402

HYD = ['R', 'S', 'G', 'H']
FIELDS = ['ZH', 'ZV', 'ZDR', 'RHOHV', 'KDP', 'ATT_H', 'ATT_V', 'DELTA_HV', 'PHIDP']


def main():
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    import _cases
    from cosmo_pol_amd import RadarOperator, gpm, synthetic
    from cosmo_pol_oracle import beam, scatter
    from cosmo_pol_oracle import config as ocfg
    from cosmo_pol_oracle import gpm as ogpm
    rng = np.random.default_rng(seed)
    cube = synthetic.small_test_cube(hydrometeors=('R', 'S', 'G'), two_moment=True, **_cases.gen_golden.CUBE_KW)
    luts = {13.6: {h: synthetic.make_lut(h, 13.6, '2mom') for h in HYD},
            35.6: {h: synthetic.make_lut(h, 35.6, '2mom') for h in HYD},
            5.6: {h: synthetic.make_lut(h, 5.6, '2mom', n_e=2, n_t=2) for h in HYD}}
    order = _cases.ORDER_2MOM
    for case in range(n_cases):
        band = 'Ku' if rng.random() < 0.6 else 'Ka'
        nv = int(rng.choice([1, 3]))
        sw = gpm.synthetic_swath(n_scans=int(rng.integers(1, 4)), n_rays=int(rng.integers(1, 6)),
                                 centre=(46.5 + rng.uniform(-0.1, 0.1), 7.5 + rng.uniform(-0.1, 0.1)),
                                 heading_deg=float(rng.uniform(0, 360)),
                                 cross_track_deg=float(rng.uniform(0.5, 5.0)),
                                 scan_spacing_m=float(rng.uniform(2000, 8000)))
        tag = 'case %d band %s nv %d swath %s' % (case, band, nv, sw['Latitude'].shape)
        try:
            base = {'radar': {'coords': [46.5, 7.5, 1000], 'frequency': 5.6, 'K_squared': 0.93},
                    'microphysics': {'scheme': '2mom', 'with_ice_crystals': 0, 'with_melting': 0,
                                     'with_attenuation': int(rng.random() < 0.7)},
                    'integration': {'nh_GH': 1, 'nv_GH': nv}}
            op = RadarOperator(config=base, luts=lambda hl, f, s: {h: luts[f][h] for h in hl},
                               output_variables='only_radar', lanes=1)
            op.load_model_arrays(cube['data'], cube['zlevels'], cube['proj_info'], cube['resolution'])
            out = op.get_GPM_swath(sw, band)
            freq, res_m = gpm.band_settings(band)
            over = {k: dict(v) for k, v in base.items()}
            over['radar'].update(frequency=freq, radial_resolution=res_m, sensitivity=12.0, type='GPM')
            over['radar']['3dB_beamwidth'] = 0.5
            conf = ocfg.make_config(over)
            ocube = beam.ModelCube({n: cube['data'][n].copy() for n in order}, cube['zlevels'],
                                   cube['proj_info'], cube['resolution'], order)
            olut = {h: _cases.as_oracle_lut(luts[freq][h]) for h in HYD}
            az, el, rg, sat = ogpm.swath_angles(sw)
            N, M = sw['Latitude'].shape
            raw = out.raw
            for idx in range(N * M):
                i, j = divmod(idx, M)
                subs, k0, n = ogpm.interpolate_swath_ray(ocube, conf, az[i, j], el[i, j], rg[i, j], sat[i])
                assert n == out.n_kept[i, j], (idx, n, out.n_kept[i, j])
                o = scatter.radar_observables(subs, olut, conf, doppler=False)
                scatter.cut_at_sensitivity([o], conf)
                assert np.array_equal(raw['mask'][idx, :n], o.mask)
                for k in FIELDS:
                    scale = np.nanmax(np.abs(o.values[k])) if np.isfinite(o.values[k]).any() else 0.0
                    _cases.assert_close_nan(raw[k][idx, :n], o.values[k], rtol=1e-5, atol=2e-5 * scale,
                                            name='%s ray %d' % (k, idx))
            op.close()
            print('ok  ', tag, flush=True)
        except Exception:
            print('FAIL', tag, flush=True)
            traceback.print_exc()
            sys.exit(1)
    print('all %d cases passed' % n_cases)


if __name__ == '__main__':
    main()
