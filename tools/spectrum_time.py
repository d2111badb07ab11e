import sys, time, json
sys.path.insert(0, '.')
import numpy as np
import bench
from cosmo_pol_amd import RadarOperator, synthetic
conf = bench.bench_config(False)
conf['doppler'] = {'scheme': 3}
hyds = ('R', 'S', 'G')
cube = synthetic.make_cube(hydrometeors=hyds, **synthetic.BENCH_GRID)
luts = synthetic.make_all_luts(hyds, 5.6, '1mom')
op = RadarOperator(config=conf, luts=luts, output_variables='only_radar', lanes=1)
op.load_model_arrays(cube['data'], cube['zlevels'], cube['proj_info'], cube['resolution'])
az = np.arange(0, 360, 1.0); el = np.full(360, 1.0)
res = op.simulate_rays(az, el)
op._ctx.enable_timing(True)
t0 = time.perf_counter()
for _ in range(3):
    res = op.simulate_rays(az, el)
dt = (time.perf_counter() - t0) / 3
c = op._ctx.counters()
print(json.dumps(dict(ms_wall=1e3 * dt, ms_final_stage=c.ms_final, ms_total=c.ms_total,
                      spectrum_shape=list(res['DSPECTRUM'].shape),
                      nonzero_bins=int(np.nansum(res['DSPECTRUM'] > 0)),
                      finite_rvel=int(np.isfinite(res['RVEL']).sum()))))
