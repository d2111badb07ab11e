"""The product's multi-process path on the device: RadarOperator(distributed=True) with two
ranks sharing the one GPU of the test box (gloo; the 8-GPU RCCL run is the driver's)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def test_two_ranks_one_gpu_bitwise_equal_to_single_process():
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', OMP_NUM_THREADS='1')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2',
           '--master-addr', '127.0.0.1', '--master-port', '29533',
           os.path.join(HERE, '_dist_gpu_worker.py')]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
    assert 'DIST_GPU_OK world=2' in out.stdout
