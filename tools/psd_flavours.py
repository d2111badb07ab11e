#!/usr/bin/env python
"""Per-flavour cost of the PSD x table stage on one sweep of the C2 / C3 / C4 workload:
items and work units per hydrometeor (bucket counters), then the PSD stage time with only one
kernel flavour launched (CPOL_PSD_ONLY, one process per mask).
   python tools/psd_flavours.py --config c3|c4|c2 [--elev 3.0] [--steps 5]"""
import argparse
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child(args):
    import numpy as np
    import bench
    from cosmo_pol_amd import RadarOperator, synthetic
    wl = args.config
    conf = bench.bench_config(False, wl)
    hyds = list(bench.hydrometeors_of(wl))
    cube = synthetic.make_cube(hydrometeors=tuple(h for h in hyds if h in 'RSGI'), **synthetic.BENCH_GRID)
    luts = synthetic.make_all_luts(hyds, 5.6, '1mom')
    op = RadarOperator(config=conf, luts=luts, output_variables='only_radar', lanes=1)
    op.load_model_arrays(cube['data'], cube['zlevels'], cube['proj_info'], cube['resolution'])
    az = np.arange(0, 360, 1.0)
    el = np.full(360, args.elev)
    out = {}
    if args.counts:
        op._ctx.enable_debug(True)
        op.simulate_rays(az, el)
        n_keys = sum(luts[h].value_table.shape[0] * luts[h].value_table.shape[1] for h in hyds)
        cnt = op._ctx.debug_read('bucket_count', (n_keys,), np.int32)
        base = 0
        for h in hyds:
            nk = luts[h].value_table.shape[0] * luts[h].value_table.shape[1]
            c = cnt[base:base + nk].astype(np.int64)
            base += nk
            ne = c[c > 0]
            out[h] = dict(items=int(c.sum()), slices=int(len(ne)), max_bucket=int(c.max()))
        op._ctx.enable_debug(False)
    op.simulate_rays(az, el)
    op._ctx.enable_timing(True)
    for _ in range(args.steps):
        op.simulate_rays(az, el)
    c = op._ctx.counters()
    out['stages_ms'] = dict(interp=c.ms_interp, classify=c.ms_classify, bucket=c.ms_bucket, psd=c.ms_psd,
                            final=c.ms_final, total=c.ms_total)
    out['n_valid'] = int(c.n_valid_items)
    print('RESULT ' + json.dumps(out), flush=True)
    op.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--config', default='c3')
    ap.add_argument('--elev', type=float, default=3.0)
    ap.add_argument('--steps', type=int, default=5)
    ap.add_argument('--counts', action='store_true')
    ap.add_argument("--child", action="store_true")
    ap.add_argument("--tag", default="")
    args = ap.parse_args()
    if args.child:
        return child(args)
    for mask, name in ((15, 'all'), (2, 'recurrence (R,S,G)'), (4, 'ice'), (8, 'melting')):
        env = dict(os.environ, CPOL_PSD_ONLY=str(mask))
        cmd = [sys.executable, os.path.abspath(__file__), '--child', '--config', args.config, '--elev',
               str(args.elev), '--steps', str(args.steps)] + (['--counts'] if mask == 15 else [])
        r = subprocess.run(cmd, env=env, capture_output=True, text=True)
        line = [l for l in r.stdout.splitlines() if l.startswith('RESULT ')]
        print(name, line[-1][7:] if line else ('FAILED ' + r.stderr[-500:]), flush=True)


if __name__ == '__main__':
    main()
