#!/usr/bin/env python
"""Where does the host time of bench.py's `host_outputs` step go, and why does it have a slow mode?

Round 4: the step as the reference's API hands a sweep over (a new elevation every step: per-ray tables computed on
the host and uploaded, kernels, all 15 arrays copied to page-locked host memory) ran at 0.255 ms per sweep in some
timed regions and at 0.53 - 0.64 ms in others of the SAME process (`host_submit_ms_per_sweep` 0.40 - 0.45 in the slow
ones).  This tool runs that step in timed regions of 20 sweeps exactly as bench.py does and splits the host time of
every region into its sections -- geometry.ray_tables (NumPy), the argument structs, PinnedPool.take, cpol_run_sweep
(the library call: staging copy, H2D, 4 launches, D2H), the rest (result views, bookkeeping) -- next to what the host
looked like: the core the thread ran on, its clock (/proc/cpuinfo), context switches, page-locked blocks allocated.

Phases: `cold` (first regions of the process), `steady`, `after_pause` (2 s of sleep first), `after_cpu_load`
(all cores busy with NumPy workers for a few seconds first, as the CPU pool legs of bench.py leave the host),
`long` (regions of 400 sweeps).

  python tools/host_mode_probe.py [--small] [--out profiles/r5_host_mode_probe.json]"""
import argparse
import json
import os
import resource
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402


def cpu_mhz(cpu):
    try:
        with open('/sys/devices/system/cpu/cpu%d/cpufreq/scaling_cur_freq' % cpu) as f:
            return int(f.read()) / 1000.0
    except (OSError, ValueError):
        pass
    try:
        n = -1
        with open('/proc/cpuinfo') as f:
            for line in f:
                if line.startswith('processor'):
                    n = int(line.split(':')[1])
                elif line.startswith('cpu MHz') and n == cpu:
                    return float(line.split(':')[1])
    except (OSError, ValueError):
        pass
    return None


def current_cpu():
    try:
        return os.sched_getcpu()
    except (AttributeError, OSError):
        return -1


class Sections(object):
    def __init__(self):
        self.t = {}

    def wrap(self, name, fn):
        def inner(*a, **kw):
            t0 = time.perf_counter()
            try:
                return fn(*a, **kw)
            finally:
                self.t[name] = self.t.get(name, 0.0) + time.perf_counter() - t0
        return inner

    def take(self):
        t, self.t = self.t, {}
        return t


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--small', action='store_true')
    ap.add_argument('--out', default=None)
    ap.add_argument('--sweeps', type=int, default=20)
    ap.add_argument('--quick', action='store_true', help='the first four regions only (e.g. under AMD_LOG_LEVEL=4)')
    args = ap.parse_args()
    import torch
    import bench
    from cosmo_pol_amd import RadarOperator, bind_to_device_numa_node, synthetic
    from cosmo_pol_amd import _native as N
    from cosmo_pol_amd import geometry as geo
    conf = bench.bench_config(args.small, 'c2')
    hyds = bench.hydrometeors_of('c2')
    if args.small:
        cube = synthetic.small_test_cube(hydrometeors=hyds)
        luts = synthetic.make_all_luts(hyds, 5.6, '1mom', n_e=8)
    else:
        cube = synthetic.make_cube(hydrometeors=hyds, **synthetic.BENCH_GRID)
        luts = synthetic.make_all_luts(hyds, 5.6, '1mom')
    numa = bind_to_device_numa_node(0)
    torch.cuda.set_device(0)
    op = RadarOperator(config=conf, luts=luts, output_variables='only_radar', device=0)
    op.load_model_arrays(cube['data'], cube['zlevels'], cube['proj_info'], cube['resolution'])
    n_lanes = 3
    for i in range(n_lanes):
        op._lane(i)
    az = np.arange(0, 360, 1.0)
    els = [np.full(len(az), e) for e in bench.C2_ELEVATIONS]
    n_gates = len(op.constants.RANGE_RADAR)
    op.reuse_device_tables = False
    sec = Sections()
    geo.ray_tables = sec.wrap('ray_tables', geo.ray_tables)
    op._pool.take = sec.wrap('pool_take', op._pool.take)
    N.Context.run_sweep = sec.wrap('cpol_run_sweep', N.Context.run_sweep)
    counter = [0]

    def step_full():
        k = counter[0]
        counter[0] += 1
        return op.simulate_rays(az, els[k % len(els)], pinned=True, lane=k % n_lanes)

    def fence():
        for i in range(n_lanes):
            op.wait(i)
        torch.cuda.synchronize()

    def region(n):
        fence()
        sec.take()
        ru0 = resource.getrusage(resource.RUSAGE_THREAD)
        c0, m0 = current_cpu(), None
        m0 = cpu_mhz(c0) if c0 >= 0 else None
        t0 = time.perf_counter()
        for _ in range(n):
            step_full()
        t_submit = time.perf_counter() - t0
        fence()
        el = time.perf_counter() - t0
        c1 = current_cpu()
        ru1 = resource.getrusage(resource.RUSAGE_THREAD)
        s = sec.take()
        known = sum(s.values())
        # the library's own split of cpol_run_sweep (host ns by section, summed over the lanes)
        lib = np.zeros(10)
        for i in range(n_lanes):
            lib += op._lane(i).debug_read('host_times', (10,), np.float64)
        calls = max(lib[0], 1.0)
        return {'ms_per_sweep': 1e3 * el / n, 'submit_ms_per_sweep': 1e3 * t_submit / n,
                'sections_us_per_sweep': dict({k: 1e6 * v / n for k, v in s.items()}, rest=1e6 * (t_submit - known) / n),
                'inside_cpol_run_sweep_us': {'tables_staging_and_h2d_call': 1e-3 * lib[1] / calls, 'work_buffers': 1e-3 * lib[2] / calls,
                                             'kernel_launches': 1e-3 * lib[3] / calls, 'd2h_copy_call': 1e-3 * lib[4] / calls,
                                             'all': 1e-3 * lib[5] / calls,
                                             'of_tables: wait_for_staging_slot': 1e-3 * lib[6] / calls, 'of_tables: fill_slot': 1e-3 * lib[7] / calls,
                                             'of_tables: hipMemcpyAsync_h2d': 1e-3 * lib[8] / calls, 'of_tables: hipEventRecord': 1e-3 * lib[9] / calls},
                'cpu': [c0, c1], 'cpu_mhz_before': m0, 'cpu_mhz_after': cpu_mhz(c1) if c1 >= 0 else None,
                'ctx_switches': [ru1.ru_nvcsw - ru0.ru_nvcsw, ru1.ru_nivcsw - ru0.ru_nivcsw],
                'minor_faults': ru1.ru_minflt - ru0.ru_minflt, 'pinned_blocks_allocated': op._pool.n_alloc}

    def show(tag, r):
        s = r['sections_us_per_sweep']
        li = r['inside_cpol_run_sweep_us']
        print('%-15s %.3f ms/sweep  submit %.3f  [ray_tables %5.0f  take %4.0f  run_sweep %5.0f (tables+h2d %4.0f = wait %3.0f fill %3.0f '
              'h2d %3.0f record %3.0f;  buffers %4.0f  launches %4.0f  d2h call %4.0f)  rest %5.0f us]  csw %s  faults %d  blocks %d'
              % (tag, r['ms_per_sweep'], r['submit_ms_per_sweep'], s.get('ray_tables', 0), s.get('pool_take', 0),
                 s.get('cpol_run_sweep', 0), li['tables_staging_and_h2d_call'], li['of_tables: wait_for_staging_slot'], li['of_tables: fill_slot'],
                 li['of_tables: hipMemcpyAsync_h2d'], li['of_tables: hipEventRecord'], li['work_buffers'], li['kernel_launches'],
                 li['d2h_copy_call'], s['rest'], r['ctx_switches'], r['minor_faults'], r['pinned_blocks_allocated']),
              file=sys.stderr, flush=True)

    rec = {'host_cpus': os.cpu_count(), 'numa': {'node': numa['node'], 'bound': numa['bound']}, 'sweeps_per_region': args.sweeps,
           'd2h_bytes_per_sweep': len(az) * n_gates * 76, 'phases': {}}

    def phase(tag, n_regions, n=args.sweeps):
        out = []
        for i in range(n_regions):
            r = region(n)
            show('%s[%d]' % (tag, i), r)
            out.append(r)
        rec['phases'][tag] = out

    if args.quick:
        print('[probe] region 0', file=sys.stderr, flush=True)
        phase('cold', 4)
        op.close()
        return
    phase('cold', 12)
    phase('steady', 8)
    time.sleep(2.0)
    phase('after_pause', 8)
    # all cores busy for a few seconds (what the CPU pool legs of bench.py do to the host), then the same regions
    n_burn = min(os.cpu_count() or 1, 256)
    burn = ('import numpy as np, time\nt = time.time()\na = np.random.rand(1024, 12, 96)\n'
            'while time.time() - t < 5.0:\n    (a * a).sum()\n')
    procs = [subprocess.Popen([sys.executable, '-c', burn], env=dict(os.environ, OMP_NUM_THREADS='1')) for _ in range(n_burn)]
    for p in procs:
        p.wait()
    phase('after_cpu_load', 12)
    phase('long', 3, n=400)
    phase('steady_again', 5)
    # the same step pinned to ONE core (what a one-process-per-GPU launcher could do), and with the thread left alone
    old = os.sched_getaffinity(0)
    try:
        os.sched_setaffinity(0, {sorted(old)[len(old) // 2]})
        phase('one_core', 6)
    finally:
        os.sched_setaffinity(0, old)
    fence()
    op.close()
    med = lambda xs: float(np.median(xs))       # noqa: E731
    rec['summary'] = {k: {'ms_per_sweep_median': med([r['ms_per_sweep'] for r in v]),
                          'ms_per_sweep_min': min(r['ms_per_sweep'] for r in v), 'ms_per_sweep_max': max(r['ms_per_sweep'] for r in v),
                          'submit_ms_per_sweep_median': med([r['submit_ms_per_sweep'] for r in v])}
                      for k, v in rec['phases'].items()}
    text = json.dumps(rec, indent=1)
    if args.out:
        with open(args.out, 'w') as f:
            f.write(text + '\n')
    print(json.dumps(rec['summary']))


if __name__ == '__main__':
    main()
