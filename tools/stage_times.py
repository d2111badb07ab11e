#!/usr/bin/env python
"""Stage times of ONE isolated sweep (one lane, HIP events around every stage), C2 / C3 / C4:
   python tools/stage_times.py --config c2 [--elev 1.0] [--steps 50]"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--config', default='c2')
    ap.add_argument('--elev', type=float, default=1.0)
    ap.add_argument('--steps', type=int, default=50)
    ap.add_argument('--tag', default='')
    ap.add_argument('--rays', type=int, default=360, help='azimuths of the sweep (1 deg apart)')
    ap.add_argument('--volume', action='store_true',
                    help='the five elevations of the c3 / c4 volume in ONE launch sequence (rays x 5 rays)')
    ap.add_argument('--blocks', type=int, default=0,
                    help='N: the contiguous ray blocks of an N-rank run one after the other (load balance of the ranks)')
    ap.add_argument('--chunk', type=int, default=0,
                    help='with --blocks: block-cyclic instead of contiguous (chunks of this many rays dealt round robin)')
    args = ap.parse_args()
    import contextlib
    import numpy as np
    import torch
    import bench
    from cosmo_pol_amd import RadarOperator, synthetic
    wl = args.config
    if wl == 'c5':
        return c5_swath(args, bench, np)
    conf = bench.bench_config(False, wl)
    hyds = list(bench.hydrometeors_of(wl))
    cube = synthetic.make_cube(hydrometeors=tuple(h for h in hyds if h in 'RSGI'), **synthetic.BENCH_GRID)
    luts = synthetic.make_all_luts(hyds, 5.6, '1mom')
    import time
    with contextlib.redirect_stdout(sys.stderr):
        t0 = time.time()
        op = RadarOperator(config=conf, luts=luts, output_variables='only_radar', lanes=1)
        t_tables = time.time() - t0
        op.load_model_arrays(cube['data'], cube['zlevels'], cube['proj_info'], cube['resolution'])
    n_gates = len(op.constants.RANGE_RADAR)
    hyds_n = len(hyds)
    if args.blocks:
        per = -(-360 // args.blocks)
        for b in range(args.blocks):
            if args.chunk:
                ids = np.arange(360)
                rays = ids[(ids // args.chunk) % args.blocks == b].astype(float)
            else:
                rays = np.arange(b * per, min(360, (b + 1) * per), 1.0)
            one_block(op, bench, np, torch, args, rays, n_gates, hyds_n,
                      wl, t_tables, 'block %d/%d chunk %d' % (b, args.blocks, args.chunk))
    else:
        one_block(op, bench, np, torch, args, np.arange(0, args.rays, 1.0), n_gates, hyds_n, wl, t_tables, args.tag)
    op.close()


def c5_swath(args, bench, np):
    """ONE Ku swath of the c5 workload (200 scans x 49 rays, 2-moment R,S,G,H,I) through get_GPM_swath."""
    import contextlib
    from cosmo_pol_amd import RadarOperator, gpm, synthetic
    hyds = bench.hydrometeors_of('c5')
    cube = synthetic.make_cube(hydrometeors=hyds, two_moment=True, **synthetic.BENCH_GRID)
    sets = {}

    def luts(hl, freq, scheme):
        if (tuple(hl), freq) not in sets:
            sets[(tuple(hl), freq)] = synthetic.make_all_luts(hl, freq, scheme)
        return sets[(tuple(hl), freq)]
    sw = gpm.synthetic_swath(n_scans=200, n_rays=49, centre=(46.5, 7.5), cross_track_deg=17.0, scan_spacing_m=3000.0)
    with contextlib.redirect_stdout(sys.stderr):
        op = RadarOperator(config=bench.bench_config(False, 'c5'), luts=luts, output_variables='only_radar', lanes=1)
        op.load_model_arrays(cube['data'], cube['zlevels'], cube['proj_info'], cube['resolution'])
        for _ in range(2):
            out = op.get_GPM_swath(sw, 'Ku')
        op._ctx.enable_timing(True)
        for _ in range(max(3, args.steps // 5)):
            op.get_GPM_swath(sw, 'Ku')
        c = op._ctx.counters()
    print(json.dumps(dict(tag=args.tag, config='c5', band='Ku', rays=int(out.azimuths.size), n_valid=int(c.n_valid_items),
                          n_table=int(c.n_table_items), n_sbg=int(c.n_subbeam_gates),
                          interp=round(c.ms_interp * 1e3, 1), classify=round(c.ms_classify * 1e3, 1),
                          bucket=round(c.ms_bucket * 1e3, 1), psd=round(c.ms_psd * 1e3, 1),
                          final=round(c.ms_final * 1e3, 1), total_us=round(c.ms_total * 1e3, 1))), flush=True)
    op.close()


def one_block(op, bench, np, torch, args, az, n_gates, n_hyd, wl, t_tables, tag):
    n_rays = len(az)
    el = np.full(n_rays, args.elev)
    if args.volume:
        az = np.tile(az, len(bench.C4_ELEVATIONS))
        el = np.repeat(np.asarray(bench.C4_ELEVATIONS, dtype=float), n_rays)
    slab = torch.empty((9, len(az), n_gates), dtype=torch.float32, device='cuda')
    ptrs = {k: slab[i].data_ptr() for i, k in enumerate(bench.RADAR_FIELDS)}
    for _ in range(3):
        op.simulate_rays(az, el, device_outputs=ptrs)
    op.wait()
    op._ctx.enable_timing(True)
    for _ in range(args.steps):
        op.simulate_rays(az, el, device_outputs=ptrs)
    op.wait()
    c = op._ctx.counters()
    chk = op._ctx.itab_report()['check']
    print(json.dumps(dict(tag=tag, config=wl, elev='volume' if args.volume else args.elev, rays=len(az), n_valid=int(c.n_valid_items),
                          n_table=int(c.n_table_items), tables_s=round(t_tables, 2),
                          itab_check=[float('%.2e' % v) for v in chk[:n_hyd]],
                          interp=round(c.ms_interp * 1e3, 1), classify=round(c.ms_classify * 1e3, 1),
                          bucket=round(c.ms_bucket * 1e3, 1), psd=round(c.ms_psd * 1e3, 1),
                          final=round(c.ms_final * 1e3, 1), total_us=round(c.ms_total * 1e3, 1))), flush=True)


if __name__ == '__main__':
    main()
