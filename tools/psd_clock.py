#!/usr/bin/env python
"""Effective shader clock inside the dominant PSD kernel on the bench workload:
s_memtime (shader cycles) over s_memrealtime (100 MHz) between the start and the end of
every persistent workgroup of the recurrence-flavour PSD kernel.  Needs the debug build:
   make -C cosmo_pol_amd/csrc clean all PROBE=1 && python tools/psd_clock.py"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402


def main():
    import bench
    from cosmo_pol_amd import RadarOperator, synthetic
    conf = bench.bench_config(False)
    hyds = ('R', 'S', 'G')
    cube = synthetic.make_cube(hydrometeors=hyds, **synthetic.BENCH_GRID)
    luts = synthetic.make_all_luts(hyds, 5.6, '1mom')
    op = RadarOperator(config=conf, luts=luts, output_variables='only_radar', lanes=1)
    op.load_model_arrays(cube['data'], cube['zlevels'], cube['proj_info'], cube['resolution'])
    az = np.arange(0, 360, 1.0)
    el = np.full(len(az), 1.0)
    op._ctx.enable_debug(True)
    out = []
    for it in range(6):
        for _ in range(5 if it else 1):
            op.simulate_rays(az, el)
        c = op._ctx.debug_read('psd_clock', (2048, 4), np.int64)
        c = c[c[:, 3] > 0]
        cyc = (c[:, 2] - c[:, 0]).astype(float)
        us = (c[:, 3] - c[:, 1]).astype(float) / 100.0
        t0 = c[:, 1].min()
        start = (c[:, 1] - t0) / 100.0
        end = (c[:, 3] - t0) / 100.0
        q = lambda a: [float(x) for x in np.percentile(a, [0, 10, 50, 90, 100])]
        out.append(dict(n_workgroups=int(len(c)), span_us=float(end.max()),
                        start_us_pct=q(start), end_us_pct=q(end), duration_us_pct=q(us),
                        ghz=float(np.median(cyc / np.maximum(us, 1e-3)) / 1e3),
                        started_after_40us=int((start > 40).sum())))
    print(json.dumps(out, indent=1))
    op.close()


if __name__ == '__main__':
    main()
