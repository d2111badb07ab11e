import sys, time, cProfile, pstats, io
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
az = np.arange(0, 360, 1.0); elevs = [0.5, 1.5, 3.0, 5.0, 8.0][:int(sys.argv[1]) if len(sys.argv) > 1 else 5]
op.lanes = int(sys.argv[2]) if len(sys.argv) > 2 else 1
op.get_PPI(elevs, azimuths=az)
pr = cProfile.Profile(); pr.enable()
t0 = time.perf_counter()
for _ in range(20): op.get_PPI(elevs, azimuths=az)
dt = (time.perf_counter() - t0) / 20
pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats('cumulative').print_stats(30); print(s.getvalue()[:3500]); print('volume ms', dt * 1e3)
