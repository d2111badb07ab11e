#!/usr/bin/env python
"""Host-side profile (cProfile) of the submission of one C4 volume (5 sweeps, device outputs):
   python tools/profile_c4_host.py [n_rays] [lanes]"""
import contextlib
import cProfile
import os
import pstats
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    n_rays = int(sys.argv[1]) if len(sys.argv) > 1 else 360
    n_lanes = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    import numpy as np
    import torch
    import bench
    from cosmo_pol_amd import RadarOperator, synthetic
    conf = bench.bench_config(False, 'c4')
    hyds = list(bench.hydrometeors_of('c4'))
    cube = synthetic.make_cube(hydrometeors=tuple(h for h in hyds if h in 'RSGI'), **synthetic.BENCH_GRID)
    luts = synthetic.make_all_luts(hyds, 5.6, '1mom')
    with contextlib.redirect_stdout(sys.stderr):
        op = RadarOperator(config=conf, luts=luts, output_variables='only_radar', lanes=n_lanes)
        op.load_model_arrays(cube['data'], cube['zlevels'], cube['proj_info'], cube['resolution'])
    az = np.arange(0, n_rays, 1.0)
    ng = len(op.constants.RANGE_RADAR)
    slabs = [torch.empty((9, n_rays, ng), dtype=torch.float32, device='cuda') for _ in range(5)]
    ptrs = [{k: s[i].data_ptr() for i, k in enumerate(bench.RADAR_FIELDS)} for s in slabs]

    def volume():
        for e, elev in enumerate(bench.C4_ELEVATIONS):
            op.simulate_rays(az, np.full(n_rays, elev), device_outputs=ptrs[e], lane=e % n_lanes)

    def wait():
        for i in range(n_lanes):
            op.wait(i)
    for _ in range(2):
        volume()
    wait()
    t0 = time.perf_counter()
    for _ in range(5):
        volume()
    t_sub = (time.perf_counter() - t0) / 5
    wait()
    t_all = (time.perf_counter() - t0) / 5
    print('lanes %d ' % n_lanes, end='')
    print('n_rays %d: host submit %.2f ms per volume, wall %.2f ms per volume' % (n_rays, 1e3 * t_sub, 1e3 * t_all))
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(5):
        volume()
    pr.disable()
    wait()
    pstats.Stats(pr, stream=sys.stdout).sort_stats('cumulative').print_stats(22)
    op.close()


if __name__ == '__main__':
    main()
