#!/usr/bin/env python
"""Measures what the SHORT form of the sub-beam geodesy changes (cpol_interp.inl, CPOL_INTERP_FAST_SUB: the
non-central sub-beams take 4 Vincenty passes, reciprocal roots by Newton, short series for atan / asin and a product
with 180 / pi) against the LONG form the central sub-beam takes (cpol_sweep_params.debug_flags =
CPOL_DEBUG_EXACT_SUBBEAMS runs every sub-beam through it), on the device, at the scale of the C4 volume:

  (a) float32 rotated coordinates (rlat, rlon) of sub-beam gates that differ, and by how many float32 ulp;
  (b) sub-beam gates whose model cell (i0, i1) = floor((rlat - La1) / dlat), floor((rlon - Lo1) / dlon)
      (interpolation_c.c:43-52) or whose mask differs -- the level index follows from the cell's columns and the
      gate height, which the two forms share: "bit-exact for gate/bin indexing" holds iff this count is 0;
  (c) the interpolated model values that differ, and the worst relative change of any output field.

  python tools/fast_sub_check.py [--rays 360] [--elevations 0.5 1.5 3 5 8] [--out profiles/r5_fast_sub_check.json]

One elevation at a time (debug reads of a 360-ray sweep with 49 sub-beams: 8.8 M sub-beam gates)."""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

FIELDS = ['ZH', 'ZV', 'ZDR', 'KDP', 'DELTA_HV', 'PHIDP', 'RHOHV', 'ATT_H', 'ATT_V', 'RVEL']
DEVICE_OUTPUTS = False          # --config c2 / c3: see compare_sweep


def cell_index(coords, llc, res):
    """(i0, i1) as gate_geometry computes them: float32 quotient, floor."""
    p0 = (coords[:, 0] - np.float32(llc[1])) / np.float32(res[1])
    p1 = (coords[:, 1] - np.float32(llc[0])) / np.float32(res[0])
    return np.floor(p0.astype(np.float64)).astype(np.int64), np.floor(p1.astype(np.float64)).astype(np.int64)


def ulp_distance(a, b):
    ia = a.view(np.int32).astype(np.int64)
    ib = b.view(np.int32).astype(np.int64)
    ia = np.where(ia < 0, -(ia & 0x7fffffff), ia)
    ib = np.where(ib < 0, -(ib & 0x7fffffff), ib)
    return np.abs(ia - ib)


def compare_sweep(op, az, el):
    """One sweep both ways -> dict of counts."""
    from cosmo_pol_amd import _native as N
    n_vars = len(op._staged_vars)
    got = {}
    for form, flags in (('short', 0), ('long', N.DEBUG_EXACT_SUBBEAMS)):
        op.debug_flags = flags
        op._ctx.enable_debug(True)
        op.simulate_rays(az, el)
        n_sbg = int(op._ctx.counters().n_subbeam_gates)
        d = {'coords': op._ctx.debug_read('sub_coords', (n_sbg, 2), np.float32),
             'mask': op._ctx.debug_read('sub_mask', (n_sbg,), np.int8),
             'vals': op._ctx.debug_read('sub_values', (n_vars, n_sbg), np.float32)}
        op._ctx.enable_debug(False)
        if DEVICE_OUTPUTS:
            # a single-beam sweep takes the polynomials only when nobody asks for its float64 latitude / longitude: results into
            # a device slab, as bench.py's c2 step does (the nine float32 fields)
            import torch
            import bench
            ng = len(op.constants.RANGE_RADAR)
            slab = torch.full((len(bench.RADAR_FIELDS), len(az), ng), float('nan'), dtype=torch.float32, device='cuda')
            op._ctx.enable_debug(True)
            op.simulate_rays(az, el, device_outputs={k: slab[i].data_ptr() for i, k in enumerate(bench.RADAR_FIELDS)})
            op.wait()
            d['coords'] = op._ctx.debug_read('sub_coords', (n_sbg, 2), np.float32)
            d['mask'] = op._ctx.debug_read('sub_mask', (n_sbg,), np.int8)
            d['vals'] = op._ctx.debug_read('sub_values', (n_vars, n_sbg), np.float32)
            op._ctx.enable_debug(False)
            op.simulate_rays(az, el, device_outputs={k: slab[i].data_ptr() for i, k in enumerate(bench.RADAR_FIELDS)})
            op.wait()
            host = slab.cpu().numpy()
            d['out'] = {k: host[i].copy() for i, k in enumerate(bench.RADAR_FIELDS)}
            d['out']['mask'] = d['mask'].copy()
        else:
            d['out'] = {k: v.copy() for k, v in op.simulate_rays(az, el).items() if k in FIELDS or k == 'mask'}
        got[form] = d
    op.debug_flags = 0
    s, l = got['short'], got['long']
    n_sbg = len(s['mask'])
    ok = np.isfinite(s['coords']).all(axis=1) & np.isfinite(l['coords']).all(axis=1)
    ulp = ulp_distance(s['coords'][ok].ravel(), l['coords'][ok].ravel())
    proj = op._proj
    llc = (float(proj['Lo1']), float(proj['La1']))
    si, sj = cell_index(s['coords'][ok], llc, op._res)
    li, lj = cell_index(l['coords'][ok], llc, op._res)
    cell_diff = int(np.sum((si != li) | (sj != lj)))
    mask_diff = int(np.sum(s['mask'] != l['mask']))
    vs, vl = s['vals'], l['vals']
    both = np.isfinite(vs) & np.isfinite(vl)
    val_diff_gates = int(np.sum(np.any((vs != vl) & both, axis=0)))
    nan_diff = int(np.sum(np.isnan(vs) != np.isnan(vl)))
    with np.errstate(divide='ignore', invalid='ignore'):
        rel = np.abs(vs[both] - vl[both]) / np.maximum(np.abs(vl[both]), 1e-30)
    worst_val = float(rel.max()) if rel.size else 0.0
    out_worst, out_diff = {}, {}
    for k in FIELDS:
        if k not in s['out']:
            out_diff[k], out_worst[k] = 0, 0.0
            continue
        a, b = s['out'][k].astype(np.float64), l['out'][k].astype(np.float64)
        fin = np.isfinite(a) & np.isfinite(b)
        out_diff[k] = int(np.sum(a[fin] != b[fin])) + int(np.sum(np.isnan(a) != np.isnan(b)))
        scale = np.maximum(np.abs(b[fin]), 1e-3 * np.max(np.abs(b[fin])) if fin.any() else 1.0)
        out_worst[k] = float(np.max(np.abs(a[fin] - b[fin]) / scale)) if fin.any() else 0.0
    return {'n_subbeam_gates': n_sbg, 'n_coordinates': int(2 * ok.sum()),
            'a_coordinates_that_differ': int(np.sum(ulp > 0)), 'a_max_ulp': int(ulp.max()) if ulp.size else 0,
            'b_cells_that_differ': cell_diff, 'b_masks_that_differ': mask_diff,
            'c_gates_with_a_model_value_that_differs': val_diff_gates, 'c_nan_pattern_differs': nan_diff,
            'c_worst_relative_change_of_a_model_value': worst_val,
            'c_output_gates_that_differ': out_diff, 'c_worst_relative_change_of_an_output': out_worst,
            'radial_mask_equal': bool(np.array_equal(s['out']['mask'], l['out']['mask']))}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--rays', type=int, default=360)
    ap.add_argument('--elevations', type=float, nargs='*', default=[0.5, 1.5, 3.0, 5.0, 8.0])
    ap.add_argument('--small', action='store_true')
    ap.add_argument('--config', default='c4', help="c4: 49 sub-beams (the non-central ones); c2 / c3: ONE sub-beam -- the polynomials of a "
                    "single-beam sweep's table set against the long form (CPOL_GEO_POLY_CENTRAL=2 keeps them under the debug reads)")
    ap.add_argument('--out', default=None)
    args = ap.parse_args()
    if args.config != 'c4':
        os.environ['CPOL_GEO_POLY_CENTRAL'] = '2'
        global DEVICE_OUTPUTS
        DEVICE_OUTPUTS = True
    import bench
    from cosmo_pol_amd import RadarOperator, synthetic
    conf = bench.bench_config(args.small, args.config)
    hyds = list(bench.hydrometeors_of(args.config))
    if args.small:
        cube = synthetic.small_test_cube(hydrometeors=('R', 'S', 'G', 'I'))
        luts = synthetic.make_all_luts(hyds, 5.6, '1mom', n_e=8)
    else:
        cube = synthetic.make_cube(hydrometeors=tuple(h for h in hyds if h in 'RSGI'), **synthetic.BENCH_GRID)
        luts = synthetic.make_all_luts(hyds, 5.6, '1mom')
    op = RadarOperator(config=conf, luts=luts, output_variables='only_radar', lanes=1)
    op.load_model_arrays(cube['data'], cube['zlevels'], cube['proj_info'], cube['resolution'])
    az = np.arange(0.0, 360.0, 360.0 / args.rays)
    per_el, tot = [], None
    for e in args.elevations:
        r = compare_sweep(op, az, np.full(len(az), e))
        r['elevation'] = e
        per_el.append(r)
        print('[fast_sub_check] el %.1f: %d of %d coordinates differ (max %d ulp), %d cells, %d masks, %d gates with a '
              'different model value' % (e, r['a_coordinates_that_differ'], r['n_coordinates'], r['a_max_ulp'],
                                         r['b_cells_that_differ'], r['b_masks_that_differ'],
                                         r['c_gates_with_a_model_value_that_differs']), file=sys.stderr, flush=True)
    op.close()
    tot = {k: sum(r[k] for r in per_el) for k in ('n_subbeam_gates', 'n_coordinates', 'a_coordinates_that_differ',
                                                   'b_cells_that_differ', 'b_masks_that_differ',
                                                   'c_gates_with_a_model_value_that_differs', 'c_nan_pattern_differs')}
    tot['a_max_ulp'] = max(r['a_max_ulp'] for r in per_el)
    tot['a_fraction_of_coordinates_that_differ'] = tot['a_coordinates_that_differ'] / max(1, tot['n_coordinates'])
    tot['c_worst_relative_change_of_a_model_value'] = max(r['c_worst_relative_change_of_a_model_value'] for r in per_el)
    tot['c_worst_relative_change_of_an_output'] = {k: max(r['c_worst_relative_change_of_an_output'][k] for r in per_el) for k in FIELDS}
    tot['c_output_gates_that_differ'] = {k: sum(r['c_output_gates_that_differ'][k] for r in per_el) for k in FIELDS}
    rec = {'what': 'short form of the sub-beam geodesy (default) against the long form (debug_flags = CPOL_DEBUG_EXACT_SUBBEAMS) '
                   'on the device: %s configuration, %d rays x %d elevations x %s x 500 gates' % (args.config, len(az), len(args.elevations),
                   '49 sub-beams' if args.config == 'c4' else 'ONE sub-beam (the polynomials of the sweep\'s table set)'),
           'total': tot, 'per_elevation': per_el}
    text = json.dumps(rec, indent=1)
    if args.out:
        with open(args.out, 'w') as f:
            f.write(text + '\n')
    print(json.dumps(tot))
    return 0 if tot['b_cells_that_differ'] == 0 and tot['b_masks_that_differ'] == 0 else 1


if __name__ == '__main__':
    sys.exit(main())
