#!/usr/bin/env python
"""Lane occupancy of k_subbeam_sum on the C4 volume (needs a library built with -DCPOL_SUBSUM_STATS:
tools/variants.sh "stats|-DCPOL_SUBSUM_STATS||python tools/subsum_stats.py"):
   python tools/subsum_stats.py [rays_per_sweep=360]"""
import contextlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
import bench  # noqa: E402
from cosmo_pol_amd import RadarOperator, synthetic  # noqa: E402

args = [a for a in sys.argv[1:] if not a.startswith('--')]
rays = int(args[0]) if args else 360
conf = bench.bench_config(False, 'c4')
hyds = list(bench.hydrometeors_of('c4'))
cube = synthetic.make_cube(hydrometeors=('R', 'S', 'G', 'I'), **synthetic.BENCH_GRID)
luts = synthetic.make_all_luts(hyds, 5.6, '1mom')
with contextlib.redirect_stdout(sys.stderr):
    op = RadarOperator(config=conf, luts=luts, output_variables='only_radar', lanes=1)
    op.load_model_arrays(cube['data'], cube['zlevels'], cube['proj_info'], cube['resolution'])
az = np.tile(np.arange(0, rays, 1.0), 5)
el = np.repeat(np.asarray(bench.C4_ELEVATIONS, dtype=float), rays)
ng = len(op.constants.RANGE_RADAR)
slab = torch.empty((9, len(az), ng), dtype=torch.float32, device='cuda')
ptrs = {k: slab[i].data_ptr() for i, k in enumerate(bench.RADAR_FIELDS)}
op.simulate_rays(az, el, device_outputs=ptrs)
op.wait()
op._ctx.debug_read('subsum_stats', (4,), np.uint64)          # (clears)
op.simulate_rays(az, el, device_outputs=ptrs)
op.wait()
st = op._ctx.debug_read('subsum_stats', (4,), np.uint64).astype(float)
c = op._ctx.counters()
print(json.dumps({'rays_per_sweep': rays, 'wave_iterations_with_work': st[0], 'lanes_with_items': st[1],
                  'fill': st[1] / (64 * st[0]), 'scalar_rounds': st[2], 'rounds_per_iteration': st[2] / st[0],
                  'iterations_skipped': st[3], 'n_table_items': int(c.n_table_items)}))
op.close()
