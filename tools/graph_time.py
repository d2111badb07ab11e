#!/usr/bin/env python
"""Single-lane sweep rate with and without the HIP-graph replay of the launch sequence
(stage timing off, device outputs): python tools/graph_time.py"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402


def main():
    import torch
    import bench
    from cosmo_pol_amd import RadarOperator, synthetic
    conf = bench.bench_config(False)
    hyds = ('R', 'S', 'G')
    cube = synthetic.make_cube(hydrometeors=hyds, **synthetic.BENCH_GRID)
    luts = synthetic.make_all_luts(hyds, 5.6, '1mom')
    az = np.arange(0, 360, 1.0)
    el = np.full(360, 1.0)
    for use_graph, torch_streams in (('0', False), ('0', True), ('1', False)):
        os.environ['CPOL_USE_GRAPH'] = use_graph
        for lanes in (1, 3):
            op = RadarOperator(config=conf, luts=luts, output_variables='only_radar', lanes=lanes)
            op.load_model_arrays(cube['data'], cube['zlevels'], cube['proj_info'], cube['resolution'])
            n_gates = len(op.constants.RANGE_RADAR)
            keep_streams = []
            if torch_streams:
                for l in range(lanes):
                    ts = torch.cuda.Stream()
                    keep_streams.append(ts)
                    op._lane(l).set_stream(ts.cuda_stream)
            slabs = [torch.empty((9, 360, n_gates), dtype=torch.float32, device='cuda') for _ in range(lanes)]
            outs = [{k: s[i].data_ptr() for i, k in enumerate(bench.OUT_FIELDS)} for s in slabs]
            for i in range(6):
                op.simulate_rays(az, el, device_outputs=outs[i % lanes], lane=i % lanes)
            torch.cuda.synchronize()
            n = 200
            t0 = time.perf_counter()
            for i in range(n):
                op.simulate_rays(az, el, device_outputs=outs[i % lanes], lane=i % lanes)
            t_sub = time.perf_counter() - t0
            for l in range(lanes):
                op._lane(l).synchronize()
            dt = time.perf_counter() - t0
            print(json.dumps(dict(graph=use_graph, torch_streams=torch_streams, lanes=lanes, ms_per_sweep=1e3 * dt / n,
                                  host_ms_per_sweep=1e3 * t_sub / n, Mgates_s=180000 * n / dt / 1e6)), flush=True)
            op.close()


if __name__ == '__main__':
    main()
