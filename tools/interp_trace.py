#!/usr/bin/env python
"""Where a wavefront of the gate-interpolation kernel spends its time (library built with -DCPOL_INTERP_TRACE):
   make -C cosmo_pol_amd/csrc clean && make -C cosmo_pol_amd/csrc EXTRA=-DCPOL_INTERP_TRACE && python tools/interp_trace.py [--config c2|c4]
Every wavefront records the 100-MHz clock -- after waiting for everything it has issued -- at: start, trajectory + grid
coordinates, topography of the four columns, column 0 bisected, neighbour columns bracketed, variables interpolated, all stored.
c2: ONE isolated sweep (k_interp_sweep, 2 880 wavefronts); c4: one elevation of the 49-sub-beam volume (k_interp_classify; the first
131 072 wavefronts; the time after 'variables' is the classification of the gate's hydrometeors + the stores)."""
import argparse
import contextlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
import bench  # noqa: E402
from cosmo_pol_amd import RadarOperator  # noqa: E402

PHASES = ('coords', 'topography', 'bisection', 'neighbours', 'variables', 'rest_and_stores')


def report(tag, tr):
    used = (tr[:, 0] > 0) & (tr[:, 6] > 0)
    t = tr[used].astype(np.int64)
    n_ok = ((tr[used, 7] >> np.uint64(40)) & np.uint64(127)).astype(np.int64)
    xcc = ((tr[used, 7] >> np.uint64(32)) & np.uint64(15)).astype(np.int64)
    base = t[:, 0].min()
    us = lambda a: a / 100.0
    # (a wavefront whose gates are all above / below the model leaves phases 3..4 empty: carry the previous stamp forward)
    for q in range(1, 7):
        t[:, q] = np.where(t[:, q] > 0, t[:, q], t[:, q - 1])
    ph = np.diff(t[:, :7], axis=1)
    life = t[:, 6] - t[:, 0]
    full = n_ok >= 48
    out = {'tag': tag, 'wavefronts': int(used.sum()), 'wavefronts_mostly_inside_the_model': int(full.sum()),
           'span_us': round(float(us(t[:, 6].max() - base)), 1), 'all_started_after_us': round(float(us(t[:, 0].max() - base)), 1),
           'sum_of_lives_us': round(float(us(life.sum())), 0)}
    for name, m in (('inside', full), ('others', ~full)):
        if m.any():
            d = {'n': int(m.sum()), 'us_life_mean': round(float(us(life[m]).mean()), 2), 'us_life_p95': round(float(np.percentile(us(life[m]), 95)), 2)}
            for k, pn in enumerate(PHASES):
                d['us_' + pn] = round(float(us(ph[m, k]).mean()), 2)
            out[name] = d
    out['by_xcc'] = {int(k): {'n': int((xcc == k).sum()), 'last_end_us': round(float(us(t[xcc == k, 6].max() - base)), 1),
                              'wave_us': round(float(us(life[xcc == k]).sum()), 0)} for k in np.unique(xcc)}
    print(json.dumps(out), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--config', default='c2')
    ap.add_argument('--tag', default='')
    args = ap.parse_args()
    wl = args.config
    conf, hyds, cube, luts = bench.make_inputs(wl, False)
    with contextlib.redirect_stdout(sys.stderr):
        op = RadarOperator(config=conf, luts=luts, output_variables='only_radar', lanes=1)
        op.load_model_arrays(cube['data'], cube['zlevels'], cube['proj_info'], cube['resolution'])
    az = np.arange(0, 360, 1.0)
    ng = len(op.constants.RANGE_RADAR)
    slab = torch.empty((9, 360, ng), dtype=torch.float32, device='cuda')
    outs = {k: slab[i].data_ptr() for i, k in enumerate(bench.RADAR_FIELDS)}
    N = 131072
    elevs = (1.0, 8.0) if wl == 'c2' else (bench.C4_ELEVATIONS[2],)
    for el in elevs:
        for _ in range(4):
            op.simulate_rays(az, np.full(360, el), device_outputs=outs)
        op.wait()
        torch.cuda.synchronize()
        report('%s %s el %.1f: one isolated sweep' % (args.tag, wl, el), op._ctx.debug_read('subsum_trace', (N, 8), np.uint64))
    op.close()


if __name__ == '__main__':
    main()
