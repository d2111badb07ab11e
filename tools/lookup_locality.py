#!/usr/bin/env python
"""How many DISTINCT integral-table blocks (LUT slice x lambda panel) do the 64 items of a
wavefront of k_psd_lookup touch, for different item -> lane tilings?
   python tools/lookup_locality.py --config c4 --elev 3"""
import argparse
import contextlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--config', default='c2')
    ap.add_argument('--elev', type=float, default=1.0)
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
    el = np.full(n_rays, args.elev)
    ng = len(op.constants.RANGE_RADAR)
    slab = torch.empty((9, n_rays, ng), dtype=torch.float32, device='cuda')
    ptrs = {k: slab[i].data_ptr() for i, k in enumerate(bench.RADAR_FIELDS)}
    op._ctx.debug_read_enable() if hasattr(op._ctx, 'debug_read_enable') else op._ctx.lib.cpol_debug_read(op._ctx.h, b'enable', None, 0)
    op.simulate_rays(az, el, device_outputs=ptrs)
    op.wait()
    c = op._ctx.counters()
    n_sbg = int(c.n_subbeam_gates)
    n_sub = n_sbg // (n_rays * ng)
    nh = len(hyds)
    key = op._ctx.debug_read('item_key', (nh, n_sbg), np.int32)
    par = op._ctx.debug_read('item_par', (nh, 6, n_sbg), np.float64)
    out = dict(config=wl, elev=args.elev, n_sub=n_sub, n_sbg=n_sbg, hydro=hyds)
    for j, h in enumerate(hyds):
        pf = par[j, 4]
        ok = (key[j] >= 0) & (pf >= 0)
        if not ok.any():
            continue
        blk = np.where(ok, key[j].astype(np.int64) * 4096 + np.floor(np.where(ok, pf, 0)).astype(np.int64), -1)
        blk = blk.reshape(n_rays, n_sub, ng)
        res = dict(items=int(ok.sum()), unique_blocks_in_sweep=int(len(np.unique(blk[blk >= 0]))))

        def distinct(tiles):     # tiles: [n_tiles, 64]
            s = np.sort(tiles, axis=1)
            d = (np.diff(s, axis=1) != 0).sum(1) + 1
            d = d - (s[:, 0] < 0)            # the -1 group is not a block
            nz = (tiles >= 0).sum(1)
            return float(d[nz > 0].mean()), float(d.sum() * 1408.0 / max(1, nz.sum()))
        g64 = ng // 64 * 64
        t = blk[:, :, :g64].reshape(-1, 64)
        res['64_gates'] = distinct(t)
        for rr, gg in ((8, 8), (4, 16), (16, 4), (2, 32)):
            r2, g2 = n_rays // rr * rr, ng // gg * gg
            t = blk[:r2, :, :g2].reshape(r2 // rr, rr, n_sub, g2 // gg, gg).transpose(0, 2, 3, 1, 4).reshape(-1, 64)
            res['%drays_x_%dgates' % (rr, gg)] = distinct(t)
        if n_sub >= 7:
            # 7 horizontal neighbours? sub-beam order unknown: try sub-blocks of 7 consecutive sub-beams x 8 gates (56 lanes)
            s2, g2 = n_sub // 7 * 7, ng // 8 * 8
            t = blk[:, :s2, :g2].reshape(n_rays, s2 // 7, 7, g2 // 8, 8).transpose(0, 1, 3, 2, 4).reshape(-1, 56)
            res['7subs_x_8gates'] = distinct(np.concatenate([t, -np.ones((t.shape[0], 8), dtype=t.dtype)], axis=1))
        out[h] = res
    print(json.dumps(out))
    op.close()


if __name__ == '__main__':
    main()
