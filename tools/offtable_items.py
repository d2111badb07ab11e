#!/usr/bin/env python
"""Lists the valid items of one sweep that are NOT on an integral table (they go to the integrating
kernels): species, parameters.   python tools/offtable_items.py --config c4 --elev 3"""
import argparse
import contextlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--config', default='c4')
    ap.add_argument('--elev', type=float, default=3.0)
    args = ap.parse_args()
    import numpy as np
    import torch
    import bench
    from cosmo_pol_amd import RadarOperator, synthetic
    wl = args.config
    conf = bench.bench_config(False, wl)
    hyds = list(bench.hydrometeors_of(wl))
    cube = synthetic.make_cube(hydrometeors=tuple(h for h in hyds if h in 'RSGI'), **synthetic.BENCH_GRID)
    luts = synthetic.make_all_luts(hyds, 5.6, '1mom')
    with contextlib.redirect_stdout(sys.stderr):
        op = RadarOperator(config=conf, luts=luts, output_variables='only_radar', lanes=1)
        op.load_model_arrays(cube['data'], cube['zlevels'], cube['proj_info'], cube['resolution'])
    n_rays = 360
    az = np.arange(0, n_rays, 1.0)
    ng = len(op.constants.RANGE_RADAR)
    slab = torch.empty((9, n_rays, ng), dtype=torch.float32, device='cuda')
    ptrs = {k: slab[i].data_ptr() for i, k in enumerate(bench.RADAR_FIELDS)}
    op._ctx.enable_debug(True)
    op.simulate_rays(az, np.full(n_rays, args.elev), device_outputs=ptrs)
    op.wait()
    c = op._ctx.counters()
    n_sbg = int(c.n_subbeam_gates)
    nh = len(hyds)
    key = op._ctx.debug_read('item_key', (nh, n_sbg), np.int32)
    par = op._ctx.debug_read('item_par', (nh, 6, n_sbg), np.float64)
    out = dict(n_valid=int(c.n_valid_items), n_table=int(c.n_table_items), off=[])
    for j, h in enumerate(hyds):
        off = np.nonzero((key[j] >= 0) & ~(par[j, 4] >= 0))[0]
        for i in off[:10]:
            out['off'].append(dict(h=h, sbg=int(i), key=int(key[j, i]), par=[float(v) for v in par[j, :5, i]]))
        out['n_off_' + h] = int(len(off))
    print(json.dumps(out))
    op.close()


if __name__ == '__main__':
    main()
