#!/usr/bin/env python
"""Host time of queueing ONE c2 sweep (device outputs), split: the Python of RadarOperator.simulate_rays, the
library call (cpol_run_sweep: uploads + kernel launches), and the device time per sweep with 1 / 3 lanes.
   python tools/submit_cost.py [--steps 300]"""
import argparse
import contextlib
import cProfile
import io
import json
import os
import pstats
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--steps', type=int, default=300)
    ap.add_argument('--profile', action='store_true')
    ap.add_argument('--tag', default='')
    ap.add_argument('--lanes', default='1,3', help='lane counts to measure')
    args = ap.parse_args()
    import numpy as np
    import torch
    import bench
    from cosmo_pol_amd import RadarOperator, synthetic
    lane_counts = [int(x) for x in args.lanes.split(',')]
    conf = bench.bench_config(False, 'c2')
    hyds = list(bench.hydrometeors_of('c2'))
    cube = synthetic.make_cube(hydrometeors=tuple(h for h in hyds if h in 'RSGI'), **synthetic.BENCH_GRID)
    luts = synthetic.make_all_luts(hyds, 5.6, '1mom')
    with contextlib.redirect_stdout(sys.stderr):
        op = RadarOperator(config=conf, luts=luts, output_variables='only_radar', lanes=max(lane_counts))
        op.load_model_arrays(cube['data'], cube['zlevels'], cube['proj_info'], cube['resolution'])
    az = np.arange(0, 360, 1.0)
    els = [np.full(len(az), 1.0 + 0.05 * k) for k in range(8)]
    n_gates = len(op.constants.RANGE_RADAR)
    dev = torch.device('cuda', 0)
    slabs = [torch.empty((len(bench.RADAR_FIELDS), len(az), n_gates), dtype=torch.float32, device=dev) for _ in range(max(lane_counts))]
    outs = [{k: sl[i].data_ptr() for i, k in enumerate(bench.RADAR_FIELDS)} for sl in slabs]

    def fence():
        for i in range(max(lane_counts)):
            op.wait(i)
        torch.cuda.synchronize()

    def run(n_lanes, n_el, n):
        for k in range(24):
            op.simulate_rays(az, els[k % n_el], device_outputs=outs[k % n_lanes], lane=k % n_lanes)
        fence()
        t0 = time.perf_counter()
        for k in range(n):
            op.simulate_rays(az, els[k % n_el], device_outputs=outs[k % n_lanes], lane=k % n_lanes)
        t1 = time.perf_counter()
        fence()
        t2 = time.perf_counter()
        return {'tag': args.tag, 'lanes': n_lanes, 'elevations': n_el, 'submit_us': 1e6 * (t1 - t0) / n, 'total_us': 1e6 * (t2 - t0) / n}
    res = [run(n, e, args.steps) for e in (1, 8) for n in lane_counts]
    for r in res:
        print(json.dumps(r))
    if args.profile:
        pr = cProfile.Profile()
        pr.enable()
        run(max(lane_counts), 8, args.steps)
        pr.disable()
        s = io.StringIO()
        pstats.Stats(pr, stream=s).sort_stats('cumulative').print_stats(18)
        print(s.getvalue())
    op.close()


if __name__ == '__main__':
    main()
