#!/usr/bin/env python
"""Where k_psd_rare spends its time (library built with -DCPOL_RARE_TRACE): the 100-MHz clock between the flavours, per workgroup.
   tools/variants.sh "rt|-DCPOL_RARE_TRACE||python tools/rare_trace.py 45" """
import contextlib, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
import bench  # noqa: E402
from cosmo_pol_amd import RadarOperator, synthetic  # noqa: E402
args = [a for a in sys.argv[1:] if not a.startswith('--')]
tag = sys.argv[sys.argv.index('--tag') + 1] if '--tag' in sys.argv else ''
args = [a for a in args if a != tag]
rays = int(args[0]) if args else 45
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
for _ in range(3):
    op.simulate_rays(az, el, device_outputs=ptrs)
    op.wait()
tr = op._ctx.debug_read('rare_trace', (64, 8), np.uint64).astype(np.int64)
c = op._ctx.counters()
op.close()
names = ('melting', 'ice_tables', 'ice', 'gamma_uniform', 'gamma_exp')
t = tr[:, :6]
d = np.diff(t, axis=1) / 100.0
busiest = int(np.argmax(t[:, 5] - t[:, 0]))
print(json.dumps({'tag': tag, 'rays_per_sweep': rays, 'n_valid': int(c.n_valid_items), 'n_table': int(c.n_table_items),
                  'workgroup_total_us_max': float((t[:, 5] - t[:, 0]).max() / 100.0), 'busiest_workgroup': busiest,
                  'phases_us_of_the_busiest': dict(zip(names, [float(x) for x in d[busiest]])),
                  'phases_us_max_over_workgroups': dict(zip(names, [float(x) for x in d.max(axis=0)])),
                  'span_us': float((t[:, 5].max() - t[:, 0].min()) / 100.0)}))
