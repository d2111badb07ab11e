"""One rank per GPU over RCCL: these tests switch themselves on when the box shows >= 2 GPUs (round-5 review, item 2) and are
skipped on the one-GPU boxes of this build.  They execute what a one-GPU box can only rehearse with one rank or over gloo: the
rooted `dist.gather` into slices of one device tensor, the all-gather, three scans pending over three lanes, a sharded GPM
swath, bench.py's c4 strong-scaling run and the driver's default N = 2 command -- each bitwise against the single process.
(The file name sorts last: under `pytest -x` every single-GPU test has run before the first of these starts.)"""
import os
import subprocess
import sys

import pytest
import torch

from test_gpu_distributed import _check_c4_strong, _run_and_read

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
# (device_count() does not initialise the GPU on this image: safe at collection time)
two_gpus = pytest.mark.skipif(torch.cuda.device_count() < 2, reason='needs >= 2 GPUs (one rank per GPU over RCCL)')


def _worker_over_rccl(n, port, *argv):
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', OMP_NUM_THREADS='1', CPOL_DIST_BACKEND='nccl', HSA_ENABLE_IPC_MODE_LEGACY='0')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(n),
           '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.join(HERE, '_dist_gpu_worker.py')] + list(argv)
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=420)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
    assert 'DIST_GPU_OK world=%d backend=nccl' % n in out.stdout


@two_gpus
def test_two_gpus_nccl_volume_and_swath_bitwise_equal_to_single_process():
    """Two ranks on two devices over RCCL: get_PPI(distributed) by all-gather, then the ROOTED gather to the last rank with three
    scans pending on three lanes, then a GPM swath -- each bitwise equal to the same scan / swath computed by one process."""
    _worker_over_rccl(2, 29551)


@pytest.mark.skipif(torch.cuda.device_count() < 3, reason='needs >= 3 GPUs')
def test_three_gpus_nccl_uneven_volume_split():
    _worker_over_rccl(3, 29553, '16')


def _bench_over_rccl(n, port, *flags, **env_extra):
    root = os.path.dirname(HERE)
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', OMP_NUM_THREADS='1', HSA_ENABLE_IPC_MODE_LEGACY='0', **env_extra)
    for k in ('CPOL_BENCH_BACKEND', 'CPOL_BENCH_ONE_DEVICE'):
        env.pop(k, None)
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(n),
           '--master-addr', '127.0.0.1', '--master-port', str(port),
           os.path.join(root, 'bench.py'), '--gpus', str(n), '--small', '--steps', '3', '--warmup', '1',
           '--cpu-seconds', '0'] + list(flags)
    return _run_and_read(cmd, env)


@two_gpus
@pytest.mark.parametrize('gather', ['all', 'root'])
def test_two_gpus_nccl_bench_c4_strong_scaling(gather):
    """`bench.py --workload c4 --gpus 2` as the driver launches it (RCCL, one rank per GPU), with the all-gather (the default
    until the rooted form has run on real ranks) and with the rooted gather: the gathered volume equals rank 0's own
    single-GPU volume bit for bit."""
    r = _bench_over_rccl(2, 29555 if gather == 'all' else 29557, '--workload', 'c4', CPOL_BENCH_C4_GATHER=gather)
    assert r['process_group_backend'] == 'nccl' and r['n_ranks_seen_by_rccl'] == 2
    assert r['collective'].startswith('all_gather_into_tensor / nccl' if gather == 'all' else 'gather(dst=0) / nccl')
    _check_c4_strong(r)


@two_gpus
def test_two_gpus_nccl_bench_default_weak_c2_line():
    """The driver's N = 2 command: the c2 step on every rank, ONE all-gather over RCCL at the end of the timed region, checked
    bit for bit; the c4 strong-scaling child of the same ranks."""
    r = _bench_over_rccl(2, 29559)
    assert r['n_gpus'] == 2 and r['scaling'] == 'weak' and r['gather_check'] is True and r['result_check'] is True
    assert r['process_group_backend'] == 'nccl' and r['n_ranks_seen_by_rccl'] == 2
    c4 = r['c4_strong_scaling']
    assert 'error' not in c4, c4
    _check_c4_strong(c4)
