#!/usr/bin/env python
"""Time line of one k_subbeam_sum launch, wavefront by wavefront (library built with -DCPOL_SUBSUM_TRACE):
   tools/variants.sh "trace|-DCPOL_SUBSUM_TRACE|CPOL_SUBSUM_COOP=0|python tools/subsum_trace.py 45"
start / end of every wavefront on the 100-MHz clock, its iterations with work and the SIMD it ran on ->
span of the launch, busy time per SIMD, longest wavefront, microseconds per iteration."""
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
N = 131072
tr = op._ctx.debug_read('subsum_trace', (N, 8), np.uint64)
op.close()
used = tr[:, 1] > 0
t0 = tr[used, 0].astype(np.int64)
t1 = tr[used, 1].astype(np.int64)
work = (tr[used, 2] & np.uint64(0xFFFFFFFF)).astype(np.int64)
wait = (tr[used, 2] >> np.uint64(32)).astype(np.int64) / 100.0      # team form: microseconds at the first barrier of its rounds
hw = tr[used, 3]
base = t0.min()
t0 = (t0 - base) / 100.0                                   # microseconds
t1 = (t1 - base) / 100.0
dur = t1 - t0
simd = ((hw >> np.uint64(32)) & np.uint64(15)) * np.uint64(1 << 16) + (hw & np.uint64(0xFFF0))     # XCC, SE / SH / CU / pipe / SIMD
ids, inv = np.unique(simd, return_inverse=True)
busy = np.zeros(len(ids))
np.add.at(busy, inv, dur)
wsum = np.zeros(len(ids))
np.add.at(wsum, inv, work)
heavy = work > 0
PHASES = ('round_setup', 'wait_for_blocks', 'chains', 'prefetch_tail_scale_store', 'barrier_1', 'ordered_sums', 'barrier_2')
phw = tr[used, 4:8]
ph = np.stack([(phw[:, i // 2] >> np.uint64(32 * (i % 2))) & np.uint64(0xFFFFFFFF) for i in range(7)], axis=1).astype(np.float64) / 100.0
order = np.argsort(-dur)[:8]
span = float(t1.max())
# wavefronts alive over time (20 samples)
ts = np.linspace(0, span, 21)[1:-1]
alive = [int(((t0 <= t) & (t1 > t)).sum()) for t in ts]
alive_work = [int(((t0 <= t) & (t1 > t) & heavy).sum()) for t in ts]
out = {'tag': tag, 'rays_per_sweep': rays, 'wavefronts': int(used.sum()), 'with_work': int(heavy.sum()),
       'span_us': span, 'iterations_with_work': int(work.sum()),
       'duration_us': {'max': float(dur.max()), 'p99': float(np.percentile(dur, 99)), 'median_with_work': float(np.median(dur[heavy])) if heavy.any() else 0.0,
                       'median_without': float(np.median(dur[~heavy])) if (~heavy).any() else 0.0},
       'us_per_iteration': {'median': float(np.median(dur[heavy] / work[heavy])), 'p10': float(np.percentile(dur[heavy] / work[heavy], 10)),
                            'p90': float(np.percentile(dur[heavy] / work[heavy], 90))},
       'simds_seen': int(len(ids)), 'simd_busy_us': {'max': float(busy.max()), 'mean': float(busy.mean()), 'min': float(busy.min())},
       'simd_iterations': {'max': int(wsum.max()), 'mean': float(wsum.mean())},
       'barrier_wait_us': {'median_with_work': float(np.median(wait[heavy])) if heavy.any() else 0.0, 'max': float(wait.max())},
       'team_phase_us_mean_with_work': {n: float(np.mean(ph[heavy, i])) for i, n in enumerate(PHASES)} if heavy.any() else {},
       'last_start_us': float(t0.max()), 'last_start_with_work_us': float(t0[heavy].max()),
       'longest': [{'start': float(t0[i]), 'end': float(t1[i]), 'work': int(work[i])} for i in order],
       'alive_at_5pct_steps': alive, 'alive_with_work': alive_work}
os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
with open(os.path.join(ROOT, 'gpurun_out', 'subsum_trace_%s.json' % (tag or 'x')), 'w') as fh:
    json.dump(out, fh)
print(json.dumps({k: out[k] for k in ('tag', 'wavefronts', 'with_work', 'span_us', 'duration_us', 'us_per_iteration', 'barrier_wait_us', 'team_phase_us_mean_with_work', 'last_start_us')}))
