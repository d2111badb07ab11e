#!/usr/bin/env python
"""Throughput of ONE rank's share of the C4 volume (N ranks -> ceil(360 / N) rays of each of the 5 sweeps,
one launch sequence) when consecutive volumes alternate over L lanes, as bench.py --workload c4 runs them:
   python tools/share_lanes.py [N=8] [lanes=1,2,3]"""
import contextlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
import bench  # noqa: E402
from cosmo_pol_amd import RadarOperator, synthetic  # noqa: E402

n_ranks = int(sys.argv[1]) if len(sys.argv) > 1 else 8
lane_counts = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else '1,2,3').split(',')]
conf = bench.bench_config(False, 'c4')
hyds = list(bench.hydrometeors_of('c4'))
cube = synthetic.make_cube(hydrometeors=('R', 'S', 'G', 'I'), **synthetic.BENCH_GRID)
luts = synthetic.make_all_luts(hyds, 5.6, '1mom')
with contextlib.redirect_stdout(sys.stderr):
    op = RadarOperator(config=conf, luts=luts, output_variables='only_radar', lanes=max(lane_counts))
    op.load_model_arrays(cube['data'], cube['zlevels'], cube['proj_info'], cube['resolution'])
rays = -(-360 // n_ranks)
az = np.tile(np.arange(0, rays, 1.0), 5)
el = np.repeat(np.asarray(bench.C4_ELEVATIONS, dtype=float), rays)
ng = len(op.constants.RANGE_RADAR)
lanes = [op._lane(i) for i in range(max(lane_counts))]
slabs = [torch.empty((len(bench.RADAR_FIELDS), len(az), ng), dtype=torch.float32, device='cuda') for _ in lanes]
ptrs = [{k: s[i].data_ptr() for i, k in enumerate(bench.RADAR_FIELDS)} for s in slabs]
for L in lane_counts:
    for k in range(2 * L):
        op.simulate_rays(az, el, device_outputs=ptrs[k % L], lane=k % L)
    for i in range(L):
        op.wait(i)
    n = 60
    t0 = time.perf_counter()
    for k in range(n):
        op.simulate_rays(az, el, device_outputs=ptrs[k % L], lane=k % L)
    for i in range(L):
        op.wait(i)
    dt = (time.perf_counter() - t0) / n
    print(json.dumps(dict(ranks=n_ranks, rays_per_sweep=rays, lanes=L, ms_per_volume_share=round(1e3 * dt, 4))), flush=True)
op.close()
