#!/usr/bin/env python
"""bench.py's c2 headline alone, quickly (experiments: tools/variants.sh "<tag>|<flags>||python tools/c2_value.py"):
three lanes, 8 elevations in turn, device outputs; prints {tag, us_per_sweep, value} (median of 5 regions of 400 sweeps)."""
import argparse
import contextlib
import json
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--tag', default='')
    ap.add_argument('--lanes', type=int, default=3)
    ap.add_argument('--sweeps', type=int, default=400)
    args = ap.parse_args()
    import numpy as np
    import torch
    import bench
    from cosmo_pol_amd import RadarOperator
    conf, hyds, cube, luts = bench.make_inputs('c2', False)
    with contextlib.redirect_stdout(sys.stderr):
        op = RadarOperator(config=conf, luts=luts, output_variables='only_radar')
        op.load_model_arrays(cube['data'], cube['zlevels'], cube['proj_info'], cube['resolution'])
    lanes = [op._lane(i) for i in range(args.lanes)]
    az = np.arange(0, 360, 1.0)
    els = [np.full(360, e) for e in bench.C2_ELEVATIONS[:8]]
    ng = len(op.constants.RANGE_RADAR)
    slabs = [torch.empty((9, 360, ng), dtype=torch.float32, device='cuda') for _ in range(args.lanes)]
    rv = [torch.empty((360, ng), dtype=torch.float64, device='cuda') for _ in range(args.lanes)]
    outs = [dict({k: s[i].data_ptr() for i, k in enumerate(bench.RADAR_FIELDS)}, RVEL=r.data_ptr()) for s, r in zip(slabs, rv)]
    k = [0]

    def sweep():
        i = k[0]
        k[0] += 1
        op.simulate_rays(az, els[i % 8], device_outputs=outs[i % args.lanes], lane=i % args.lanes)

    def fence():
        for i in range(args.lanes):
            op.wait(i)
        torch.cuda.synchronize()
    for _ in range(64):
        sweep()
    fence()
    t, ts = [], []
    for _ in range(5):
        t0 = time.perf_counter()
        for _ in range(args.sweeps):
            sweep()
        ts.append((time.perf_counter() - t0) / args.sweeps)
        fence()
        t.append((time.perf_counter() - t0) / args.sweeps)
    us = 1e6 * statistics.median(t)
    print(json.dumps({'tag': args.tag, 'lanes': args.lanes, 'us_per_sweep': round(us, 2), 'value': round(180000 / us * 1e6 / 1e9, 3),
                      'host_submit_us_per_sweep': round(1e6 * statistics.median(ts), 2), 'all_us': [round(1e6 * x, 2) for x in t]}))
    op.close()


if __name__ == '__main__':
    main()
