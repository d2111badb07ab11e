#!/usr/bin/env python
"""Host-side profile (cProfile) of get_GPM_swath on the 2-moment bench cube (config C5):
   python tools/profile_c5_host.py [Ku|Ka]"""
import cProfile
import os
import pstats
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    band = sys.argv[1] if len(sys.argv) > 1 else 'Ku'
    import bench
    from cosmo_pol_amd import RadarOperator, gpm, synthetic
    conf = bench.bench_config(False)
    conf['radar']['type'] = 'GPM'
    conf['microphysics'].update(scheme='2mom', with_melting=0, with_ice_crystals=1)
    cube = synthetic.make_cube(hydrometeors=('R', 'S', 'G', 'H', 'I'), two_moment=True, **synthetic.BENCH_GRID)
    luts = lambda hl, freq, scheme: synthetic.make_all_luts(hl, freq, scheme)   # noqa: E731
    op = RadarOperator(config=conf, luts=luts, output_variables='only_radar')
    op.load_model_arrays(cube['data'], cube['zlevels'], cube['proj_info'], cube['resolution'])
    n_rays = 49 if band == 'Ku' else 25
    swath = gpm.synthetic_swath(n_scans=200, n_rays=n_rays, centre=(46.5, 7.5),
                                cross_track_deg=17.0 if band == 'Ku' else 8.5, scan_spacing_m=3000.0)
    op.get_GPM_swath(swath, band)
    op.get_GPM_swath(swath, band)
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(3):
        op.get_GPM_swath(swath, band)
    pr.disable()
    st = pstats.Stats(pr, stream=sys.stdout)
    st.sort_stats('cumulative').print_stats(28)
    op.close()


if __name__ == '__main__':
    main()
