"""Worker of tests/test_gpu_distributed.py: two ranks on ONE GPU (gloo rendezvous; RCCL
rejects two ranks on the same device) run RadarOperator(distributed=True).get_PPI and
compare with the same scan computed locally by each rank."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'oracle'), os.path.join(ROOT, 'tests')):
    sys.path.insert(0, p)

import numpy as np  # noqa: E402
import torch.distributed as dist  # noqa: E402


def main():
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
        op = RadarOperator(config=over, luts=luts, output_variables='only_radar', device=0,
                           distributed=distributed)
        op.load_model_arrays(cube['data'], cube['zlevels'], cube['proj_info'], cube['resolution'])
        scans.append(op.get_PPI(elevations=[4.0, 6.0], azimuths=azs))
        op.close()
    a, b = scans
    for k in b.fields:
        if k not in a.fields:
            continue
        x, y = np.ma.asarray(a.fields[k]['data']), np.ma.asarray(b.fields[k]['data'])
        assert x.shape == y.shape, k
        assert np.array_equal(np.ma.getmaskarray(x), np.ma.getmaskarray(y)), k
        assert np.array_equal(x.filled(0), y.filled(0)), (k, rank)
    dist.barrier()
    if rank == 0:
        print('DIST_GPU_OK world=%d fields=%d' % (world, len(a.fields)))
    dist.destroy_process_group()


if __name__ == '__main__':
    main()
