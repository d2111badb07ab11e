"""Worker of tests/test_gpu_distributed.py: two or three ranks on ONE GPU (gloo rendezvous; RCCL
rejects two ranks on the same device) -- or, with CPOL_DIST_BACKEND=nccl on a box with >= 2 GPUs, one rank per GPU over
RCCL (the rooted `dist.gather` on device tensors, three scans pending: what a one-GPU box cannot execute) -- run RadarOperator(distributed=True).get_PPI (rays of every
sweep sharded) and .get_GPM_swath (scan lines sharded) and compare with the same scan / swath
computed locally by each rank, bit for bit."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'oracle'), os.path.join(ROOT, 'tests')):
    sys.path.insert(0, p)

import numpy as np  # noqa: E402
import torch.distributed as dist  # noqa: E402


def main():
    backend = os.environ.get('CPOL_DIST_BACKEND', 'gloo')
    local = int(os.environ.get('LOCAL_RANK', '0')) if backend == 'nccl' else 0      # (gloo: every rank on GPU 0)
    if backend == 'nccl':
        import torch
        torch.cuda.set_device(local)
        dist.init_process_group('nccl', device_id=torch.device('cuda', local))
    else:
        dist.init_process_group('gloo')
    rank, world = dist.get_rank(), dist.get_world_size()
    import _cases
    from cosmo_pol_amd import RadarOperator
    name = 'c4_subbeams'
    over = _cases.gen_golden.radial_case_inputs(name)[0]
    over = {k: dict(v) for k, v in over.items()}
    _, _, _, _, luts, cube = _cases.radial_case(name)
    n_az = int(sys.argv[1]) if len(sys.argv) > 1 else 15
    azs = np.arange(n_az) * (360. / n_az)       # 15 rays: uneven split over 2 ranks; 16 over 3: 6 + 6 + 4
    scans = []
    for distributed in (True, False):
        op = RadarOperator(config=over, luts=luts, output_variables='only_radar', device=local,
                           distributed=distributed)
        op.load_model_arrays(cube['data'], cube['zlevels'], cube['proj_info'], cube['resolution'])
        scans.append(op.get_PPI(elevations=[4.0, 6.0], azimuths=azs))
        op.close()
    a, b = scans
    # the rooted form: rank 1 alone receives, assembles and copies the scan; pipelined submissions on
    # alternating lanes (collective + copy of scan k beside the kernels of scan k + 1) equal the blocking call
    op = RadarOperator(config=over, luts=luts, output_variables='only_radar', device=local, distributed=True,
                       gather_to=world - 1, lanes=3)
    op.load_model_arrays(cube['data'], cube['zlevels'], cube['proj_info'], cube['resolution'])
    rooted = op.get_PPI(elevations=[4.0, 6.0], azimuths=azs)
    sweeps = [[(azs, np.full(len(azs), e)) for e in els] for els in ([4.0, 6.0], [5.0, 7.0], [4.0, 6.0])]
    pend = [op.submit_volume(sw, fields=['ZH', 'KDP', 'mask'], lane=i % 3) for i, sw in enumerate(sweeps)]   # three scans pending
    got = [q.wait() for q in pend]
    for i in range(3):
        op.wait(i)
    # (gloo has no rooted gather of device tensors: all-gather there, and still only the root assembles and copies)
    assert op._dist_runner().collective.startswith('gather(dst=%d) / nccl' % (world - 1) if backend == 'nccl'
                                                   else 'all_gather_into_tensor / gloo')
    if rank == world - 1:
        # torch (through the HIP runtime) sees the operator's page-locked result block as pinned: copy_(non_blocking=True) into it
        # is an asynchronous DMA on the runner's stream, not a staged pageable copy
        assert op._dist_runner().host_pinned is True
    if rank == world - 1:
        for k in b.fields:
            if k in rooted.fields:
                x, y = np.ma.asarray(rooted.fields[k]['data']), np.ma.asarray(b.fields[k]['data'])
                assert np.array_equal(np.ma.getmaskarray(x), np.ma.getmaskarray(y)) and np.array_equal(x.filled(0), y.filled(0)), k
        assert all(g is not None and sorted(g[0]) == ['KDP', 'ZH', 'mask'] for g in got)
        for s in range(2):
            for k in ('ZH', 'KDP', 'mask'):
                assert np.array_equal(got[0][s][k], got[2][s][k], equal_nan=True)      # the same scan twice
                ref = np.asarray(b.raw[s]['fields'][k] if k != 'mask' else b.raw[s]['mask'])
                assert np.array_equal(got[0][s][k], ref, equal_nan=True), (k, s)
        assert not np.array_equal(got[0][0]['ZH'], got[1][0]['ZH'], equal_nan=True)    # another elevation
    else:
        assert rooted is None and all(g is None for g in got)
    op.close()
    for k in b.fields:
        if k not in a.fields:
            continue
        x, y = np.ma.asarray(a.fields[k]['data']), np.ma.asarray(b.fields[k]['data'])
        assert x.shape == y.shape, k
        assert np.array_equal(np.ma.getmaskarray(x), np.ma.getmaskarray(y)), k
        assert np.array_equal(x.filled(0), y.filled(0)), (k, rank)
    # a GPM swath (BASELINE configs[4]): scan lines sharded over the ranks, every rank gets the whole swath
    from cosmo_pol_amd import gpm, synthetic
    cube2 = synthetic.small_test_cube(hydrometeors=('R', 'S', 'G'), two_moment=True, **_cases.gen_golden.CUBE_KW)
    base = {'radar': {'coords': [46.5, 7.5, 1000], 'frequency': 5.6, 'K_squared': 0.93},
            'microphysics': {'scheme': '2mom', 'with_ice_crystals': 0, 'with_melting': 0},
            'integration': {'nh_GH': 1, 'nv_GH': 3}}
    tabs = {}

    def provider(hl, freq, scheme):
        for h in hl:
            if (h, freq) not in tabs:
                tabs[(h, freq)] = synthetic.make_lut(h, freq, '2mom', n_e=2, n_t=2)
        return {h: tabs[(h, freq)] for h in hl}
    sw = gpm.synthetic_swath(n_scans=7, n_rays=5, cross_track_deg=4.0, scan_spacing_m=6000.0)   # 7 lines: uneven split
    swaths = []
    for distributed in (True, False):
        op = RadarOperator(config=base, luts=provider, output_variables='only_radar', device=local,
                           distributed=distributed)
        op.load_model_arrays(cube2['data'], cube2['zlevels'], cube2['proj_info'], cube2['resolution'])
        swaths.append(op.get_GPM_swath(sw, 'Ku'))
        op.close()
    sa, sb = swaths
    assert np.array_equal(sa.bin_surface, sb.bin_surface)
    assert np.array_equal(sa.lats, sb.lats, equal_nan=True) and np.array_equal(sa.lons, sb.lons, equal_nan=True)
    n_fin = 0
    for k in sb.data:
        assert np.array_equal(sa.data[k], sb.data[k], equal_nan=True), (k, rank)
        n_fin += int(np.isfinite(sb.data[k]).sum())
    assert n_fin > 100
    dist.barrier()
    if rank == 0:
        print('DIST_GPU_OK world=%d backend=%s fields=%d swath_fields=%d' % (world, backend, len(a.fields), len(sb.data)))
    dist.destroy_process_group()


if __name__ == '__main__':
    main()
