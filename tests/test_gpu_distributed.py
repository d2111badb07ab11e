"""The product's multi-process path on the device: RadarOperator(distributed=True) with two
ranks sharing the one GPU of the test box (gloo; the 8-GPU RCCL run is the driver's).

The tests that need two GPUs (one rank per GPU over RCCL) live in tests/test_gpu_zz_two_gpus.py: collected last."""
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


def test_three_ranks_one_gpu_uneven_volume_split():
    """Three ranks (15 rays: 5 + 5 + 5 per sweep; gloo) through the same worker: the volume layout with
    more than two ranks."""
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', OMP_NUM_THREADS='1')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '3',
           '--master-addr', '127.0.0.1', '--master-port', '29537',
           os.path.join(HERE, '_dist_gpu_worker.py'), '16']
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
    assert 'DIST_GPU_OK world=3' in out.stdout


def test_one_rank_nccl_worker_rooted_gather_three_scans_pending():
    """The worker of the two-GPU tests (tests/test_gpu_zz_two_gpus.py) in its RCCL mode with ONE rank: process group with a
    device id, the rooted `dist.gather` over `nccl` with three scans pending on three lanes, the page-locked result block,
    the sharded swath -- what a one-GPU box can execute of that code."""
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', OMP_NUM_THREADS='1', CPOL_DIST_BACKEND='nccl', HSA_ENABLE_IPC_MODE_LEGACY='0')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '1',
           '--master-addr', '127.0.0.1', '--master-port', '29539', os.path.join(HERE, '_dist_gpu_worker.py')]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
    assert 'DIST_GPU_OK world=1 backend=nccl' in out.stdout


def _bench_two_ranks(port, *flags):
    import json
    root = os.path.dirname(HERE)
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', OMP_NUM_THREADS='1', CPOL_BENCH_BACKEND='gloo',
               CPOL_BENCH_ONE_DEVICE='1')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2',
           '--master-addr', '127.0.0.1', '--master-port', str(port),
           os.path.join(root, 'bench.py'), '--gpus', '2', '--small', '--steps', '2', '--warmup', '1',
           '--cpu-seconds', '0'] + list(flags)
    return _run_and_read(cmd, env)


def _run_and_read(cmd, env):
    """Runs bench.py; checks that stdout ends with ONE compact JSON line (< 4 KB, the contract's keys) preceded only by
    '#detail ' lines, and returns the FULL result (bench_detail.json, written where CPOL_BENCH_DETAIL says)."""
    import json
    import tempfile
    with tempfile.TemporaryDirectory() as tmp:
        detail = os.path.join(tmp, 'detail.json')
        out = subprocess.run(cmd, env=dict(env, CPOL_BENCH_DETAIL=detail), capture_output=True, text=True, timeout=420)
        assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
        rows = out.stdout.splitlines()
        lines = [l for l in rows if l.startswith('{')]
        assert len(lines) == 1 and rows[-1] == lines[0]          # rank 0 alone prints; the compact line is the last one
        assert len(lines[0]) < 4096
        assert any(l.startswith('#detail ') for l in rows[:-1])      # (the full result also travels on prefixed lines)
        line = json.loads(lines[0])
        with open(detail) as f:
            full = json.load(f)
    for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'scaling', 'dtype', 'data', 'config', 'roofline'):
        assert k in line, k
        if k not in ('config', 'roofline', 'value', 'ms_per_step'):
            assert line[k] == full[k], k
    assert abs(line['value'] - full['value']) <= 1e-5 * full['value']
    return full


def _check_c4_strong(r):
    assert r['n_gpus'] == 2 and r['scaling'] == 'strong' and r['gather_check'] is True
    assert len(r['per_rank']) == 2
    assert r['per_rank'][0]['rays_per_sweep'] + r['per_rank'][1]['rays_per_sweep'] == 90
    assert r['single_gpu_same_workload']['value'] > 0 and r['value'] > 0 and r['speedup_vs_single_gpu'] > 0
    # both speedups compare the distributed path on N ranks with the SAME path on a one-rank group (rank 0 alone)
    one = r['single_gpu_same_workload']['through_the_distributed_path']
    assert one['pipelined_ms_per_volume'] > 0 and one['single_volume_ms'] > 0 and r['speedup_single_volume'] > 0
    assert abs(r['speedup_vs_single_gpu'] - one['pipelined_ms_per_volume'] / r['ms_per_step']) < 1e-6 * r['speedup_vs_single_gpu']
    assert r['single_volume_blocking']['ms_per_volume'] > 0
    assert r['collectives_in_timed_region'] == r['steps'] and r['n_ranks_seen_by_backend'] == 2
    assert r['api_ms']['get_PPI_distributed_median'] > 0


def test_bench_c4_strong_scaling_mode_two_ranks_one_gpu():
    """bench.py --workload c4 with two ranks (gloo, both on GPU 0): azimuths of every sweep
    sharded, one all-gather per volume, the gathered volume equals rank 0's own single-GPU
    volume bit for bit (`gather_check`)."""
    r = _bench_two_ranks(29541, '--workload', 'c4')
    assert r['config']['workload'].startswith('c4')
    _check_c4_strong(r)


def test_bench_default_two_ranks_weak_c2_line_with_c4_extra():
    """The driver's N > 1 command (default workload): the c2 step of N = 1 on every rank (weak scaling,
    rank r on another elevation), one all-gather at the end of the timed region checked bit for bit,
    and the c4 strong-scaling run of the same ranks as `c4_strong_scaling`."""
    r = _bench_two_ranks(29543)
    assert r['n_gpus'] == 2 and r['scaling'] == 'weak' and r['gather_check'] is True
    assert r['config']['workload'].startswith('c2') and r['metric'] == 'range-gates/sec'
    assert 'left in HBM' in r['config']['workload'] and r['value'] > 0     # (inputs and outputs resident: the contract's `value`)
    assert r['roofline']['frac'] is None or r['roofline']['frac'] > 0
    c4 = r['c4_strong_scaling']
    assert 'error' not in c4, c4
    assert r['c4_speedup_vs_single_gpu'] == c4['speedup_vs_single_gpu'] and r['c4_gather_check'] is True
    assert r['process_group_backend'] == 'gloo' and r['n_ranks_seen_by_rccl'] is None
    assert c4['workload'].startswith('c4')
    _check_c4_strong(c4)


def _bench_one_rank(*flags, **env_extra):
    """bench.py on ONE rank with the real backend (nccl = RCCL): what a one-GPU box can execute of the
    multi-GPU code -- process group, communicator, collectives on the side stream."""
    import json
    root = os.path.dirname(HERE)
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', OMP_NUM_THREADS='1', CPOL_BENCH_NO_EXTRAS='1', **env_extra)
    env.pop('CPOL_BENCH_BACKEND', None)
    cmd = [sys.executable, os.path.join(root, 'bench.py'), '--gpus', '1', '--small', '--steps', '3', '--warmup', '1',
           '--cpu-seconds', '0'] + list(flags)
    return _run_and_read(cmd, env)


def test_bench_c4_one_rank_through_rccl():
    """`bench.py --workload c4` at N = 1 runs the product's distributed path (RadarOperator.submit_volume)
    over a one-rank RCCL group: rooted gather on the runner's stream, device-side assembly, the blocking
    get_PPI(distributed) call; the gathered volume equals the plain single-GPU volume bit for bit."""
    r = _bench_one_rank('--workload', 'c4', CPOL_BENCH_C4_GATHER='root')
    assert r['process_group_backend'] == 'nccl' and r['n_ranks_seen_by_rccl'] == 1
    assert r['collective'].startswith('gather(dst=0) / nccl') and r['collectives_in_timed_region'] == 3
    assert r['gather_check'] is True and r['scaling'] == 'strong' and r['value'] > 0
    assert r['api_ms']['get_PPI_distributed_median'] > 0
    # the default since round 6: the all-gather (the rooted form stays opt-in until it has run on >= 2 real ranks)
    r = _bench_one_rank('--workload', 'c4')
    assert r['collective'].startswith('all_gather_into_tensor / nccl') and r['gather_check'] is True
    assert r['host_block_pinned'] is True


def test_bench_c2_weak_path_one_rank_through_rccl():
    """The N > 1 code path of the default workload (process group after the lanes, all-gather of the last
    sweep on a side stream, bitwise gather check) with one rank and RCCL (CPOL_BENCH_FORCE_COLLECTIVES)."""
    r = _bench_one_rank(CPOL_BENCH_FORCE_COLLECTIVES='1')
    assert r['process_group_backend'] == 'nccl' and r['n_ranks_seen_by_rccl'] == 1
    assert r['gather_check'] is True and r['scaling'] == 'weak' and r['value'] > 0
