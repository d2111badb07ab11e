#!/usr/bin/env python
"""Host-side profile (cProfile) of the c3 bench step: five pinned sweeps through the C ABI on five lanes."""
import cProfile
import io
import os
import pstats
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import bench  # noqa: E402
from cosmo_pol_amd import RadarOperator, synthetic  # noqa: E402

conf = bench.bench_config(False, 'c3')
hyds = bench.hydrometeors_of('c3')
cube = synthetic.make_cube(hydrometeors=('R', 'S', 'G', 'I'), **synthetic.BENCH_GRID)
luts = synthetic.make_all_luts(hyds, 5.6, '1mom')
op = RadarOperator(config=conf, luts=luts, output_variables='only_radar')
op.load_model_arrays(cube['data'], cube['zlevels'], cube['proj_info'], cube['resolution'])
az = np.arange(0, 360, 1.0)
els = [np.full(360, e) for e in bench.C4_ELEVATIONS]
op.reuse_device_tables = False


def volume():
    return [op.simulate_rays(az, els[e], pinned=True, lane=e) for e in range(5)]


for _ in range(3):
    volume()
for i in range(5):
    op.wait(i)
pr = cProfile.Profile()
pr.enable()
t0 = time.perf_counter()
for _ in range(20):
    volume()
t_sub = (time.perf_counter() - t0) / 20
for i in range(5):
    op.wait(i)
dt = (time.perf_counter() - t0) / 20
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats('tottime').print_stats(22)
print(s.getvalue()[:4000])
print('volume ms %.3f  submit ms %.3f' % (dt * 1e3, t_sub * 1e3))
op.close()
