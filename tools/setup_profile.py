#!/usr/bin/env python
"""Where the set-up time of a RadarOperator goes (tables to HBM, integral tables, model cube): cProfile of the constructor and
of load_model_arrays, plus the library's own build times (`itab_times`).
   python tools/setup_profile.py --config c4 [--repeat 2]"""
import argparse
import cProfile
import io
import os
import pstats
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--config', default='c4')
    ap.add_argument('--repeat', type=int, default=2)
    args = ap.parse_args()
    import contextlib
    import numpy as np
    import bench
    from cosmo_pol_amd import RadarOperator
    conf, hyds, cube, luts = bench.make_inputs(args.config, False)
    for r in range(args.repeat):
        pr = cProfile.Profile()
        t0 = time.time()
        with contextlib.redirect_stdout(sys.stderr):
            pr.enable()
            op = RadarOperator(config=conf, luts=luts, output_variables='only_radar')
            op._ctx.synchronize()
            pr.disable()
        t1 = time.time()
        with contextlib.redirect_stdout(sys.stderr):
            op.load_model_arrays(cube['data'], cube['zlevels'], cube['proj_info'], cube['resolution'])
            op._ctx.synchronize()
        t2 = time.time()
        s = io.StringIO()
        pstats.Stats(pr, stream=s).sort_stats('cumulative').print_stats(18)
        print('== construction %d: tables %.3f s, model cube %.3f s' % (r, t1 - t0, t2 - t1))
        print('\n'.join(l for l in s.getvalue().splitlines() if l.strip())[:4000])
        try:
            rep = op._ctx.itab_report()
            print('itab_report:', {k: (v if not isinstance(v, (list, tuple)) else list(v)) for k, v in rep.items()} if isinstance(rep, dict) else rep)
        except Exception as e:
            print('itab_report failed:', e)
        op.close()


if __name__ == '__main__':
    main()
