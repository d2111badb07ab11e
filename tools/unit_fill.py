import sys, json
sys.path.insert(0, '.')
import numpy as np
import bench
from cosmo_pol_amd import RadarOperator, synthetic
conf = bench.bench_config(False)
conf['microphysics'].update(with_melting=1, with_ice_crystals=1)
hyds = ['R', 'S', 'G', 'mS', 'mG', 'I']
cube = synthetic.make_cube(hydrometeors=('R', 'S', 'G', 'I'), **synthetic.BENCH_GRID)
luts = synthetic.make_all_luts(hyds, 5.6, '1mom')
op = RadarOperator(config=conf, luts=luts, output_variables='only_radar', lanes=1)
op.load_model_arrays(cube['data'], cube['zlevels'], cube['proj_info'], cube['resolution'])
op._ctx.enable_debug(True)
az = np.arange(0, 360, 1.0)
for e in (1.5, 5.0):
    op.simulate_rays(az, np.full(360, e))
    n_keys = sum(l.value_table.shape[0] * l.value_table.shape[1] for l in [luts[h] for h in ['R', 'S', 'G', 'mS', 'mG', 'I']])
    cnt = op._ctx.debug_read('bucket_count', (n_keys,), np.int32)
    base = 0
    out = {}
    for h in ['R', 'S', 'G', 'mS', 'mG', 'I']:
        nk = luts[h].value_table.shape[0] * luts[h].value_table.shape[1]
        c = cnt[base:base + nk]; base += nk
        ne = c[c > 0]
        units64 = int(np.sum(-(-ne // 64)))
        out[h] = dict(items=int(c.sum()), slices=int(len(ne)), units64=units64,
                      fill=float(c.sum() / max(1, units64 * 64)))
    print(e, json.dumps(out))
