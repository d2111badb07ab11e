#!/usr/bin/env python
"""Times the BASELINE configs C3 / C4 (and variants) on one GPU:
   python tools/run_config.py --config c3|c4 [--steps K]
C3: 5-elevation volume (360 x 500 each), R,S,G,mS,mG,I (melting + ice), 1 sub-beam.
C4: C3 with the 7 x 7 Gauss-Hermite antenna quadrature (49 sub-beams)."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--config', default='c3')
    ap.add_argument('--steps', type=int, default=3)
    ap.add_argument('--small', action='store_true')
    args = ap.parse_args()
    import bench
    from cosmo_pol_amd import RadarOperator, synthetic
    if args.config == 'c5':
        return run_c5(args)
    conf = bench.bench_config(args.small)
    conf['microphysics'].update(with_melting=1, with_ice_crystals=1)
    if args.config == 'c4':
        conf['integration'].update(nh_GH=7, nv_GH=7, weight_threshold=1.)
    hyds = ['R', 'S', 'G', 'mS', 'mG', 'I']
    t0 = time.time()
    if args.small:
        cube = synthetic.small_test_cube(hydrometeors=('R', 'S', 'G', 'I'))
        luts = synthetic.make_all_luts(hyds, 5.6, '1mom', n_e=8)
    else:
        cube = synthetic.make_cube(hydrometeors=('R', 'S', 'G', 'I'), **synthetic.BENCH_GRID)
        luts = synthetic.make_all_luts(hyds, 5.6, '1mom')
    print('inputs %.1f s' % (time.time() - t0), flush=True)
    op = RadarOperator(config=conf, luts=luts, output_variables='only_radar')
    op.load_model_arrays(cube['data'], cube['zlevels'], cube['proj_info'], cube['resolution'])
    az = np.arange(0, 360, 1.0)
    elevs = [0.5, 1.5, 3.0, 5.0, 8.0]
    op._ctx.enable_timing(True)
    out = []
    for e in elevs:
        el = np.full(len(az), e)
        op.simulate_rays(az, el)                       # warm-up (allocations)
        op._ctx.enable_timing(True)
        t0 = time.perf_counter()
        for _ in range(args.steps):
            res = op.simulate_rays(az, el)
        dt = (time.perf_counter() - t0) / args.steps
        c = op._ctx.counters()
        rec = dict(elevation=e, ms_wall=1e3 * dt, n_sbg=int(c.n_subbeam_gates),
                   n_valid=int(c.n_valid_items), units=int(c.n_work_units),
                   finite_zh=int(np.isfinite(res['ZH']).sum()),
                   stages_ms=dict(traj=c.ms_traj, interp=c.ms_interp, classify=c.ms_classify,
                                  bucket=c.ms_bucket, psd=c.ms_psd, final=c.ms_final,
                                  total=c.ms_total))
        print(json.dumps(rec), flush=True)
        out.append(rec)
    tot = sum(r['ms_wall'] for r in out)
    gates = len(elevs) * len(az) * res['ZH'].shape[1]
    print(json.dumps(dict(config=args.config, volume_ms=tot, gates=gates,
                          gates_per_s=gates / (tot * 1e-3))))
    # the product path: get_PPI spreads the sweeps over lanes (host threads + forked contexts)
    op._ctx.enable_timing(False)
    for lanes in (1, 3):
        op.lanes = lanes
        op.get_PPI(elevs, azimuths=az)
        t0 = time.perf_counter()
        for _ in range(args.steps):
            op.get_PPI(elevs, azimuths=az)
        dt = (time.perf_counter() - t0) / args.steps
        print(json.dumps(dict(config=args.config, api='get_PPI', lanes=lanes, volume_ms=1e3 * dt,
                              gates_per_s=gates / dt)))
    op.close()


def run_c5(args):
    """C5: GPM-DPR Ku (200 x 49 rays) and Ka (200 x 25) swaths over the 2-moment bench cube."""
    import bench
    from cosmo_pol_amd import RadarOperator, gpm, synthetic
    conf = bench.bench_config(False)
    conf['radar']['type'] = 'GPM'
    conf['microphysics'].update(scheme='2mom', with_melting=0, with_ice_crystals=1)
    hyds = ['R', 'S', 'G', 'H', 'I']
    t0 = time.time()
    cube = synthetic.make_cube(hydrometeors=('R', 'S', 'G', 'H', 'I'), two_moment=True,
                               **synthetic.BENCH_GRID)
    luts = lambda hl, freq, scheme: synthetic.make_all_luts(hl, freq, scheme)   # noqa: E731
    print('inputs %.1f s' % (time.time() - t0), flush=True)
    op = RadarOperator(config=conf, luts=luts, output_variables='only_radar')
    op.load_model_arrays(cube['data'], cube['zlevels'], cube['proj_info'], cube['resolution'])
    for band, n_rays in (('Ku', 49), ('Ka', 25)):
        swath = gpm.synthetic_swath(n_scans=200, n_rays=n_rays, centre=(46.5, 7.5),
                                    cross_track_deg=17.0 if band == 'Ku' else 8.5, scan_spacing_m=3000.0)
        t0 = time.perf_counter()
        out = op.get_GPM_swath(swath, band)            # includes the table reload of the band
        t_first = time.perf_counter() - t0
        t0 = time.perf_counter()
        for _ in range(args.steps):
            out = op.get_GPM_swath(swath, band)
        dt = (time.perf_counter() - t0) / args.steps
        n_gates = int(np.isfinite(out.lats).sum())
        print(json.dumps(dict(config='c5', band=band, rays=200 * n_rays, kept_gates=n_gates,
                              first_call_s=t_first, swath_ms=1e3 * dt, gates_per_s=n_gates / dt,
                              finite_zh=int(np.isfinite(out.data['ZH']).sum()))), flush=True)
    op.close()


if __name__ == '__main__':
    main()
