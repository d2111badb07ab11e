#!/usr/bin/env python
"""Where a wavefront of k_gate1_ray spends its time (library built with -DCPOL_SUBSUM_TRACE):
   tools/variants.sh "g1trace|-DCPOL_SUBSUM_TRACE||python tools/gate1_trace.py"
Every wavefront (gate tile, ray, species) records the 100-MHz clock at: start, model values arrived, PSD parameters done,
gather + Horner done, terms in LDS + ticket taken, and -- the wavefront that took the last ticket -- gates finished.
ONE isolated c2 sweep on one lane (CPOL_GATE1_RAY=1), and one of three lanes in flight."""
import contextlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault('CPOL_GATE1_RAY', '1')
import numpy as np  # noqa: E402
import torch  # noqa: E402
import bench  # noqa: E402
from cosmo_pol_amd import RadarOperator  # noqa: E402


def report(tag, tr):
    used = tr[:, 0] > 0
    t = tr[used].astype(np.int64)
    nv = (tr[used, 6] & np.uint64(0xFFFFFFFF)).astype(np.int64)
    sp = (tr[used, 6] >> np.uint64(32)).astype(np.int64)
    base = t[:, 0].min()
    us = lambda a: a / 100.0
    ph = np.stack([t[:, 1] - t[:, 0], t[:, 2] - t[:, 1], t[:, 3] - t[:, 2], t[:, 4] - t[:, 3]], axis=1)
    last = t[:, 5] > 0
    fin = np.where(last, t[:, 5] - t[:, 4], 0)
    life = np.where(last, t[:, 5], t[:, 4]) - t[:, 0]
    heavy = nv > 0
    out = {'tag': tag, 'wavefronts': int(used.sum()), 'with_items': int(heavy.sum()),
           'span_us': round(us(np.where(last, t[:, 5], t[:, 4]).max() - base), 1),
           'all_started_after_us': round(us(t[:, 0].max() - base), 1)}
    for name, m in (('with_items', heavy), ('empty', ~heavy)):
        if m.any():
            out[name] = {'n': int(m.sum()), 'mean_valid_lanes': round(float(nv[m].mean()), 1),
                         'us_values': round(float(us(ph[m, 0]).mean()), 2), 'us_parameters': round(float(us(ph[m, 1]).mean()), 2),
                         'us_gather_horner': round(float(us(ph[m, 2]).mean()), 2), 'us_lds_ticket': round(float(us(ph[m, 3]).mean()), 2),
                         'us_life_mean': round(float(us(life[m]).mean()), 2), 'us_life_p95': round(float(np.percentile(us(life[m]), 95)), 2)}
    if last.any():
        out['finishing'] = {'n': int(last.sum()), 'us_finish_mean': round(float(us(fin[last]).mean()), 2)}
    out['by_species_us_life'] = {int(k): round(float(us(life[(sp == k) & heavy]).mean()), 2) for k in np.unique(sp) if ((sp == k) & heavy).any()}
    # per gate tile (blockIdx.x = the workgroup's linear index mod 8): how much work it holds, when its last wavefront ends, and
    # on which XCD (XCC_ID of the hardware register) its wavefronts ran
    idx = np.nonzero(used)[0]
    tile = (idx // 3) % 8
    end = np.where(last, t[:, 5], t[:, 4])
    xcc = ((tr[used, 7] >> np.uint64(32)) & np.uint64(15)).astype(np.int64)
    out['by_tile'] = {int(k): {'with_items': int((heavy & (tile == k)).sum()), 'wave_us': round(float(us(life[tile == k]).sum()), 0),
                               'last_end_us': round(float(us(end[tile == k].max() - base)), 1),
                               'xcc_ids': sorted(set(xcc[tile == k].tolist()))} for k in range(8)}
    print(json.dumps(out), flush=True)


def main():
    conf, hyds, cube, luts = bench.make_inputs('c2', False)
    with contextlib.redirect_stdout(sys.stderr):
        op = RadarOperator(config=conf, luts=luts, output_variables='only_radar')
        op.load_model_arrays(cube['data'], cube['zlevels'], cube['proj_info'], cube['resolution'])
    lanes = [op._lane(i) for i in range(3)]
    az = np.arange(0, 360, 1.0)
    els = [np.full(360, e) for e in bench.C2_ELEVATIONS[:8]]
    ng = len(op.constants.RANGE_RADAR)
    slabs = [torch.empty((9, 360, ng), dtype=torch.float32, device='cuda') for _ in range(3)]
    outs = [{k: s[i].data_ptr() for i, k in enumerate(bench.RADAR_FIELDS)} for s in slabs]
    N = 131072
    for _ in range(4):
        op.simulate_rays(az, els[0], device_outputs=outs[0], lane=0)
    op.wait(0)
    torch.cuda.synchronize()
    report('isolated sweep', op._ctx.debug_read('subsum_trace', (N, 8), np.uint64))
    for k in range(48):
        op.simulate_rays(az, els[k % 8], device_outputs=outs[k % 3], lane=k % 3)
    for i in range(3):
        op.wait(i)
    torch.cuda.synchronize()
    report('a sweep among three lanes in flight (the last writer of every slot)', op._ctx.debug_read('subsum_trace', (N, 8), np.uint64))
    op.close()


if __name__ == '__main__':
    main()
