#!/usr/bin/env python
"""bench.py -- range-gates/s of the cosmo_pol hot path on MI355X.

  python bench.py --gpus N --steps K --warmup W
  (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

Workload (BASELINE.json configs[1]): 360-azimuth x 500-gate C-band PPI at 1.0 deg
elevation, rain + snow + graupel 1-moment, 1 sub-beam, attenuation on, on the
synthetic COSMO-1-like cube (80 x 774 x 1158) with full-size
synthetic scattering tables.  A "step" = one complete sweep through the C ABI
(all kernels, outputs left in HBM; the per-ray tables of the unchanged scan
geometry stay resident in HBM between steps -- `value_fresh_tables` re-uploads
them every step).  Consecutive steps run on alternating LANES (cpol_fork: shared
cube and tables, own stream and work buffers), so two or three sweeps are in
flight together, as the sweeps of a volume scan are in the product; stage times
and the roofline are measured on lane 0 under that overlap.  With N GPUs every rank
simulates one such sweep per step (rays sharded by whole sweeps, weak scaling)
and the output slabs are collected with ONE RCCL all-gather per step.

Prints ONE JSON line (rank 0).  `roofline` is measured live with HIP events on
the library's stream over the timed region (cpol_enable_timing / cpol_counters: two
events per sweep around the PSD stage of lane 0; all stages in the single-lane pass after it);
`cpu_baseline` times the CPU oracle (the restatement of the reference
algorithm, per radial, un-batched) on a bounded azimuth sample on this host.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, 'oracle')):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402

OUT_FIELDS = ['ZH', 'ZV', 'ZDR', 'KDP', 'DELTA_HV', 'PHIDP', 'RHOHV', 'ATT_H', 'ATT_V']
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s
LUT_SLICE_BYTES = 1024 * 12 * 4   # B_l per valid item (float32 staging figure)
# HBM bytes per sweep of the PSD kernel from the PMC passes committed under profiles/
# ((2*FETCH_SIZE + WRITE_SIZE)*1024, gfx950 correction of MI355X_MICROARCH.md); refreshed
# whenever the profile is re-taken -- see profiles/README.md
PSD_TRAFFIC_BYTES_PER_SWEEP = 66.9e6     # profiles/r1_final2_pmc_hbm.json, k_psd_uniform<false>


def bench_config(small):
    rng = 30000 if small else 150000
    return {'radar': {'coords': [46.5, 7.5, 1000], 'frequency': 5.6, 'range': rng,
                      'radial_resolution': 300, '3dB_beamwidth': 1., 'K_squared': 0.93,
                      'type': 'ground', 'sensitivity': [-5, 10000]},
            'refraction': {'scheme': 1},
            'integration': {'scheme': 1, 'nh_GH': 1, 'nv_GH': 1, 'weight_threshold': 1.},
            'doppler': {'scheme': 1},
            'microphysics': {'scheme': '1mom', 'with_melting': 0, 'with_ice_crystals': 0,
                             'with_attenuation': 1}}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=200)
    ap.add_argument('--warmup', type=int, default=10)
    ap.add_argument('--small', action='store_true', help='small cube / tables (debugging)')
    ap.add_argument('--cpu-seconds', type=float, default=15.0,
                    help='budget of the CPU-oracle baseline sample (0 = skip)')
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit('launch with torch.distributed.run for --gpus > 1')
        args.gpus = world
    # debugging aids for a one-GPU box: CPOL_BENCH_BACKEND=gloo CPOL_BENCH_ONE_DEVICE=1 runs the
    # N-rank code path with every rank on GPU 0 (never used by the driver)
    backend = os.environ.get('CPOL_BENCH_BACKEND', 'nccl')
    if os.environ.get('CPOL_BENCH_ONE_DEVICE'):
        local_rank = 0
    from cosmo_pol_amd import RadarOperator, synthetic
    conf = bench_config(args.small)
    hyds = ('R', 'S', 'G')
    t0 = time.time()
    if args.small:
        cube = synthetic.small_test_cube(hydrometeors=hyds)
        luts = synthetic.make_all_luts(hyds, 5.6, '1mom', n_e=8)
    else:
        cube = synthetic.make_cube(hydrometeors=hyds, **synthetic.BENCH_GRID)
        luts = synthetic.make_all_luts(hyds, 5.6, '1mom')
    t_gen = time.time() - t0

    # CPU baselines first: the all-core leg forks workers, which must happen before this
    # process initialises the GPU (HIP state does not survive a fork)
    cpu_res = None
    if world == 1 and args.cpu_seconds > 0:
        az_cpu = np.arange(0, 360, 1.0)
        cpu_res = cpu_baseline(conf, cube, luts, az_cpu, args.cpu_seconds)
        cpu_res['all_cores'] = cpu_baseline_pool(conf, cube, luts, az_cpu)

    torch.cuda.set_device(local_rank)
    t0 = time.time()
    import contextlib
    with contextlib.redirect_stdout(sys.stderr):      # the operator's notices: stdout carries ONE JSON line
        op = RadarOperator(config=conf, luts=luts, output_variables='only_radar', device=local_rank)
        op.load_model_arrays(cube['data'], cube['zlevels'], cube['proj_info'], cube['resolution'])
    op._ctx.synchronize()
    t_stage = time.time() - t0

    az = np.arange(0, 360, 1.0)
    el = np.full(len(az), 1.0)
    n_rays, n_gates = len(az), len(op.constants.RANGE_RADAR)
    # lanes: contexts forked from the operator's (shared cube / tables, own stream and work
    # buffers); consecutive steps go to alternating lanes so that the latency-bound kernels
    # of one sweep overlap the PSD kernel of the other (CPOL_BENCH_LANES=1 disables).  The
    # lanes keep the library's own non-blocking streams, created back to back before any other
    # stream of the process so that each gets its own hardware queue (measured: torch-created
    # streams, or forking after other streams exist, cost 6-25 % through queue sharing).
    n_lanes = max(1, int(os.environ.get('CPOL_BENCH_LANES', '3')))
    lanes = [op._lane(i) for i in range(n_lanes)]
    # the process group comes AFTER the lanes so that RCCL's own streams do not take the
    # hardware queues of the lanes
    if world > 1:
        torch.cuda.set_device(local_rank)
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if backend == 'nccl':
            dist.init_process_group('nccl', rank=rank, world_size=world,
                                    device_id=torch.device('cuda', local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    lane_streams = ([torch.cuda.ExternalStream(c.stream_ptr(), device=torch.device('cuda', local_rank))
                     for c in lanes] if world > 1 else None)
    # two output slabs: the all-gather of step i (side stream) overlaps the kernels of
    # step i+1 (library stream); a slab is reused only after its gather has completed
    n_buf = max(2, n_lanes) if world > 1 else n_lanes
    slabs = [torch.empty((len(OUT_FIELDS), n_rays, n_gates), dtype=torch.float32, device='cuda')
             for _ in range(n_buf)]
    slab = slabs[0]
    gathered = ([torch.empty(world * slab.numel(), dtype=torch.float32, device='cuda')
                 for _ in range(n_buf)] if world > 1 else None)
    dev_outs = [{k: sl[i].data_ptr() for i, k in enumerate(OUT_FIELDS)} for sl in slabs]
    dev_out = dev_outs[0]
    comm_stream = torch.cuda.Stream() if world > 1 else None
    slab_free = [None] * n_buf
    counter = [0]

    def step():
        b = counter[0] % n_buf
        lane = counter[0] % n_lanes
        counter[0] += 1
        if world == 1:
            op.simulate_rays(az, el, device_outputs=dev_outs[b], lane=lane)
            return
        if slab_free[b] is not None:
            lane_streams[lane].wait_event(slab_free[b])
        op.simulate_rays(az, el, device_outputs=dev_outs[b], lane=lane)
        computed = torch.cuda.Event()
        computed.record(lane_streams[lane])
        comm_stream.wait_event(computed)
        with torch.cuda.stream(comm_stream):
            dist.all_gather_into_tensor(gathered[b], slabs[b].view(-1))
            slab_free[b] = torch.cuda.Event()
            slab_free[b].record(comm_stream)

    def fence():
        for c in lanes:
            c.synchronize()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    if world > 1:
        # communicator set-up (lazy in RCCL) belongs to the setup phase, not to a step
        with torch.cuda.stream(comm_stream):
            dist.all_gather_into_tensor(gathered[0], slabs[0].view(-1))
        fence()
    for _ in range(args.warmup):
        step()
    fence()
    op._ctx.enable_timing(2)             # HIP events around the PSD stage of lane 0 (2 per sweep)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    t_submit = time.perf_counter() - t0      # host time to enqueue all steps
    fence()
    elapsed = time.perf_counter() - t0
    cnt = op._ctx.counters()             # also surfaces a domain error, if any
    for i in range(1, n_lanes):
        op._lane(i).counters()
    op._ctx.enable_timing(False)
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device='cuda')
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    # outside the timed region: every rank simulated the same sweep, so every gathered
    # block of the last step must equal this rank's own slab, bit for bit
    gather_ok = None
    if world > 1:
        b = (counter[0] - 1) % n_buf
        own = slabs[b].view(-1)
        blocks = gathered[b].view(world, -1)
        same = [bool(torch.equal(torch.nan_to_num(blocks[r]), torch.nan_to_num(own))) for r in range(world)]
        gather_ok = all(same)

    gates_per_step = world * n_rays * n_gates
    value = gates_per_step * args.steps / elapsed

    # the same sweep on ONE lane (no other sweep in flight): kernel durations in isolation
    iso = None
    if world == 1:
        n_it = max(5, args.steps // 2)
        op._ctx.enable_timing(True)
        for _ in range(n_it):
            op.simulate_rays(az, el, device_outputs=dev_outs[0], lane=0)
        fence()
        iso = op._ctx.counters()
        op._ctx.enable_timing(False)

    # variant that re-uploads the per-ray tables on every step (new scan geometry each time)
    value_fresh = None
    if world == 1:
        op.reuse_device_tables = False
        n_it = max(3, args.steps // 2)
        step()
        fence()
        t0 = time.perf_counter()
        for _ in range(n_it):
            step()
        fence()
        value_fresh = n_rays * n_gates * n_it / (time.perf_counter() - t0)
        op.reuse_device_tables = True

    # PCIe-inclusive variant (outputs copied to host buffers every step), N = 1 only
    value_d2h = None
    if world == 1:
        op.simulate_rays(az, el)
        t0 = time.perf_counter()
        for _ in range(max(3, args.steps // 4)):
            op.simulate_rays(az, el)
        value_d2h = n_rays * n_gates * max(3, args.steps // 4) / (time.perf_counter() - t0)

    out = None
    if rank == 0:
        n_valid = int(cnt.n_valid_items)
        n_sbg = int(cnt.n_subbeam_gates)
        n_vars = len(op._staged_vars)
        psd_bytes = n_valid * LUT_SLICE_BYTES
        achieved = psd_bytes / (cnt.ms_psd * 1e-3) / 1e9 if cnt.ms_psd > 0 else None
        sweep_bytes = (n_sbg * (4 * cube['zlevels'].shape[0] * 4 + n_vars * 8 * 4)
                       + psd_bytes + n_rays * n_gates * 48)
        # f64 VALU issue roofline of the PSD kernel: 16 v_*_f64 per (item, bin) in the
        # recurrence flavour (12 FMAs + 4, counted in the ISA), 4 cycles per wave64
        # instruction on a SIMD-32, 1024 SIMDs at the 2.4 GHz peak clock
        n_units = int(cnt.n_work_units)
        valu_cycles = -(-int(cnt.n_valid_items) // 64) * 1024 * 16 * 4
        valu_frac = valu_cycles / (1024 * 2.4e9 * cnt.ms_psd * 1e-3) if cnt.ms_psd > 0 else None
        out = {
            'metric': 'range-gates/sec', 'value': value, 'unit': 'gates/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': 1e3 * elapsed / args.steps, 'higher_is_better': True,
            'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f64', 'data': 'synthetic',
            'config': {'workload': '360-azimuth x 500-gate C-band PPI (el 1.0 deg), rain+snow+graupel '
                                   '1-moment, 1 sub-beam, synthetic %s cube; one such sweep per GPU '
                                   'per step' % ('x'.join(map(str, cube['zlevels'].shape))),
                       'rays_per_gpu': n_rays, 'gates_per_ray': n_gates,
                       'lanes': n_lanes,
                       'parallelism': ('rays sharded by sweep, 1 all-gather/step on a side stream, overlapped '
                                       'with the next step') if world > 1 else 'single GPU',
                       'small': bool(args.small)},
            'roofline': {'kernel': 'k_psd_uniform<false> (recurrence flavour of the PSD x table kernel; 1 launch/sweep covers R, S, G)', 'bound': 'hbm',
                         'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                         'frac': (achieved / HBM_PEAK_GBS) if achieved else None,
                         'traffic': PSD_TRAFFIC_BYTES_PER_SWEEP,
                         'algorithmic_bytes_per_sweep_stage': psd_bytes,
                         'valu_f64_frac': valu_frac,
                         'avg_stage_ms': cnt.ms_psd,
                         'isolated': None if iso is None else {
                             'avg_stage_ms': iso.ms_psd,
                             'frac': psd_bytes / (iso.ms_psd * 1e-3) / 1e9 / HBM_PEAK_GBS,
                             'valu_f64_frac': valu_cycles / (1024 * 2.4e9 * iso.ms_psd * 1e-3),
                             'device_total_ms': iso.ms_total,
                             'stages_ms': {'trajectory': iso.ms_traj, 'interp': iso.ms_interp,
                                           'classify': iso.ms_classify, 'bucket': iso.ms_bucket,
                                           'psd': iso.ms_psd, 'final': iso.ms_final},
                             'note': 'same sweep with one lane only (no overlap with other sweeps)'},
                         'note': 'algorithmic bytes = N_valid x 49152 B (one float32 LUT slice per valid '
                                 'item, SURVEY 8(d)); slices are shared through the scalar cache / L2, '
                                 'so frac can exceed 1 -- see DESIGN.md'},
            'stages_ms': None if iso is None else {
                'trajectory': iso.ms_traj, 'interp': iso.ms_interp, 'classify': iso.ms_classify,
                'bucket': iso.ms_bucket, 'psd': iso.ms_psd, 'final': iso.ms_final,
                'device_total': iso.ms_total, 'psd_with_lanes_in_flight': cnt.ms_psd,
                'note': 'one sweep on one lane (the pass after the timed region); in the timed '
                        'region only the PSD stage of lane 0 carries events (2 per sweep)'},
            'counters': {'n_subbeam_gates': n_sbg, 'n_valid_items': n_valid,
                         'n_work_units': int(cnt.n_work_units),
                         'sweep_algorithmic_bytes': sweep_bytes,
                         'sweep_algorithmic_GBs': sweep_bytes * world / (elapsed / args.steps) / 1e9},
            'host_submit_ms_per_step': 1e3 * t_submit / args.steps,
            'gather_check': gather_ok,
            'value_with_d2h': value_d2h,
            'value_fresh_tables': value_fresh,
            'setup_s': {'synthetic_inputs': t_gen, 'stage_to_hbm': t_stage},
        }
        if cpu_res is not None:
            out['cpu_baseline'] = cpu_res
            out['gpu_over_cpu_core'] = value / cpu_res['value']
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    op.close()


_POOL_STATE = {}


def _oracle_inputs(conf, cube, luts):
    from cosmo_pol_oracle import beam
    from cosmo_pol_oracle import config as ocfg
    from cosmo_pol_oracle import lut as olut
    oconf = ocfg.make_config(conf)
    order = ['U', 'V', 'W', 'QR_v', 'QS_v', 'QG_v', 'QI_v', 'RHO', 'T']
    oc = beam.ModelCube({n: cube['data'][n] for n in order}, cube['zlevels'], cube['proj_info'],
                        cube['resolution'], order)
    ol = {}
    for h, s in luts.items():
        L = olut.LookupTable()
        L.axes, L.axes_names, L.axes_limits, L.axes_step = s.axes, s.axes_names, s.axes_limits, s.axes_step
        L.value_table = s.value_table
        ol[h] = L
    return oconf, oc, ol


def _pool_radial(a):
    from cosmo_pol_oracle import beam, scatter
    oconf, oc, ol = _POOL_STATE['inputs']
    subs = beam.interpolate_radial(oc, oconf, float(a), 1.0)
    return len(scatter.radar_observables(subs, ol, oconf).values['ZH'])


def cpu_baseline_pool(conf, cube, luts, az, max_procs=16):
    """the same per-radial oracle under a fork pool mapped over the
    azimuths of whole sweeps.  Two legs: persistent workers (the favourable one for the
    CPU, reported as `value`) and one task per worker process as the reference's
    `Pool(processes=P, maxtasksperchild=1).map` (radar_operator.py:402,431)."""
    import multiprocessing as mp
    procs = max(1, min(os.cpu_count() or 1, max_procs))
    _POOL_STATE['inputs'] = _oracle_inputs(conf, cube, luts)     # inherited by fork, not pickled
    ctx = mp.get_context('fork')
    res = {}
    for leg, kw, sweeps in (('persistent', {}, 4), ('reference_style', {'maxtasksperchild': 1}, 1)):
        t0 = time.perf_counter()
        with ctx.Pool(processes=procs, **kw) as pool:
            n_gates = sum(pool.map(_pool_radial, list(az) * sweeps, chunksize=1))
        dt = time.perf_counter() - t0
        res[leg] = (n_gates / dt, dt, sweeps)
    _POOL_STATE.clear()
    v, dt, sweeps = res['persistent']
    return {'value': v, 'unit': 'gates/s', 'cores': procs,
            'sample': '%d sweeps of %d radials, fork pool of %d persistent worker processes, %.1f s '
                      '(pool start-up included)' % (sweeps, len(az), procs, dt),
            'reference_style': {'value': res['reference_style'][0],
                                'sample': 'one sweep, Pool(%d, maxtasksperchild=1) as '
                                          'radar_operator.py:402 (a fork per radial), %.1f s'
                                          % (procs, res['reference_style'][1])}}


def cpu_baseline(conf, cube, luts, az, budget_s):
    """The CPU oracle (restatement of the reference algorithm: per radial,
    per-variable C gate kernel, float64 LUT gather + einsum) on 1 core."""
    from cosmo_pol_oracle import beam, scatter
    from cosmo_pol_oracle import config as ocfg
    from cosmo_pol_oracle import lut as olut
    oconf = ocfg.make_config(conf)
    order = ['U', 'V', 'W', 'QR_v', 'QS_v', 'QG_v', 'QI_v', 'RHO', 'T']
    oc = beam.ModelCube({n: cube['data'][n] for n in order}, cube['zlevels'], cube['proj_info'],
                        cube['resolution'], order)
    ol = {}
    for h, s in luts.items():
        L = olut.LookupTable()
        L.axes, L.axes_names, L.axes_limits, L.axes_step = s.axes, s.axes_names, s.axes_limits, s.axes_step
        L.value_table = s.value_table
        ol[h] = L
    # whole PPIs, azimuth by azimuth like the reference's pool.map tasks, until the
    # budget is used (at least one full sweep when it fits, never less than 8 radials)
    n_done, n_gates = 0, 0
    t0 = time.perf_counter()
    done = False
    while not done:
        for a in az:
            subs = beam.interpolate_radial(oc, oconf, float(a), 1.0)
            obs = scatter.radar_observables(subs, ol, oconf)
            n_done += 1
            n_gates += len(obs.values['ZH'])
            if time.perf_counter() - t0 > budget_s and n_done >= 8:
                done = True
                break
    dt = time.perf_counter() - t0
    return {'value': n_gates / dt, 'unit': 'gates/s', 'cores': 1, 'kind': 'port',
            'sample': '%d radials (%.2f sweeps of 360 azimuths, in azimuth order), %d gates each, '
                      'oracle/cosmo_pol_oracle on one host core, %.1f s'
                      % (n_done, n_done / 360.0, n_gates // max(n_done, 1), dt),
            'host_cpus': os.cpu_count()}


if __name__ == '__main__':
    os.environ.setdefault('OMP_NUM_THREADS', '1')
    main()
