import sys, os, time, contextlib
sys.path.insert(0, os.getcwd())
import numpy as np, torch, bench
from cosmo_pol_amd import RadarOperator
conf, hyds, cube, luts = bench.make_inputs('c3', False)
with contextlib.redirect_stdout(sys.stderr):
    op = RadarOperator(config=conf, luts=luts, output_variables='only_radar')
    op.load_model_arrays(cube['data'], cube['zlevels'], cube['proj_info'], cube['resolution'])
az = np.arange(0, 360, 1.0)
els = [np.full(360, e) for e in bench.C4_ELEVATIONS]
for i in range(5): op._lane(i)
def volume(): return [op.simulate_rays(az, els[e], pinned=True, lane=e) for e in range(5)]
def fence():
    for i in range(5): op.wait(i)
    torch.cuda.synchronize()
out=[]
for r in range(16):
    t0=time.perf_counter()
    for _ in range(10): volume()
    ts=time.perf_counter()-t0
    fence()
    ht = sum(op._lane(i).debug_read('host_times', (10,), np.float64) for i in range(5))
    out.append((round(1e2*(time.perf_counter()-t0),3), round(1e2*ts,3), 'lib us/sweep: tables %.0f buffers %.0f launch %.0f copies %.0f total %.0f' % tuple(ht[k] / max(ht[0], 1) / 1e3 for k in (1, 2, 3, 4, 5))))
print('ms per volume (total, submit) per region of 10:'); [print(o) for o in out]
import subprocess
print(subprocess.run(['rocm-smi','--showclocks'],capture_output=True,text=True).stdout[-900:])
