"""N > 1 path on CPU: world_size-2 (and 3) gloo processes exercise the ray
sharding and the single all-gather of cosmo_pol_amd.distributed with a CPU
stand-in for the per-rank simulation (the HIP path itself needs a GPU)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FIELDS = ['ZH', 'ZDR', 'KDP', 'lats']
DTYPES = [np.float32, np.float32, np.float32, np.float64]
N_GATES = 37


def fake_simulate(az, el):
    """Deterministic function of (azimuth, elevation, gate) with NaN holes."""
    g = np.arange(N_GATES, dtype=np.float64)[None, :]
    base = np.sin(np.deg2rad(az))[:, None] * 100 + el[:, None] * 7 + g
    out = {}
    for i, k in enumerate(FIELDS):
        v = (base * (i + 1)).astype(DTYPES[i])
        v[:, (i + 3)::11] = np.nan
        out[k] = v
    return out


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_rays, q):
    sys.path.insert(0, ROOT)
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from cosmo_pol_amd import distributed as D
    az = np.linspace(0, 359, n_rays)
    el = np.full(n_rays, 1.5)
    calls = []

    def sim(a, e):
        calls.append(len(a))
        return fake_simulate(a, e)
    res = D.simulate_sharded(sim, az, el, list(zip(FIELDS, DTYPES)), N_GATES)
    lo, hi, per = D.shard_bounds(n_rays, world, rank)
    q.put((rank, {k: v.copy() for k, v in res.items()}, calls, (lo, hi, per)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize('world,n_rays', [(2, 360), (2, 361), (3, 10), (2, 1)])
def test_sharded_sweep_equals_single_process(world, n_rays):
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_rays, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    az = np.linspace(0, 359, n_rays)
    ref = fake_simulate(az, np.full(n_rays, 1.5))
    covered = 0
    for rank, res, calls, (lo, hi, per) in got:
        for k in FIELDS:
            assert res[k].shape == (n_rays, N_GATES)
            # bitwise identical to the un-sharded result on every rank
            assert res[k].dtype == ref[k].dtype
            assert np.array_equal(res[k].view(np.uint8), ref[k].view(np.uint8)), (rank, k)
        assert calls == ([hi - lo] if hi > lo else [])
        covered += hi - lo
    assert covered == n_rays


def test_shard_bounds_cover_and_are_contiguous():
    from cosmo_pol_amd.distributed import shard_bounds
    for n in (1, 7, 360, 361, 1800):
        for w in (1, 2, 4, 8):
            edges = [shard_bounds(n, w, r) for r in range(w)]
            assert edges[0][0] == 0 and edges[-1][1] == n
            for a, b in zip(edges[:-1], edges[1:]):
                assert a[1] == b[0] and a[2] == b[2]
            assert all(hi - lo <= per for lo, hi, per in edges)


def _worker_volume(rank, world, port, rays_per_sweep, q):
    sys.path.insert(0, ROOT)
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from cosmo_pol_amd import distributed as D
    sweeps = [(np.linspace(0, 359, n), np.full(n, 0.5 + 1.5 * s)) for s, n in enumerate(rays_per_sweep)]
    calls = []

    def sim(a, e):
        calls.append(len(a))
        return fake_simulate(a, e)
    res = D.simulate_sharded_volume(sim, sweeps, list(zip(FIELDS, DTYPES)), N_GATES)
    q.put((rank, [{k: v.copy() for k, v in r.items()} for r in res], calls))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize('world,rays', [(2, (360, 360, 360)), (3, (10, 7, 1, 4)), (2, (1, 1))])
def test_sharded_volume_one_call_one_gather_equals_single_process(world, rays):
    """A scan of several sweeps sharded as a whole: every rank runs its rays of ALL sweeps in one
    call and one all-gather assembles every sweep on every rank, bitwise equal to the un-sharded
    sweeps (the bench's c4 mode and RadarOperator(distributed=True).get_PPI)."""
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_volume, args=(r, world, port, rays, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    from cosmo_pol_amd.distributed import shard_bounds
    for rank, res, calls in got:
        mine = sum(shard_bounds(n, world, rank)[1] - shard_bounds(n, world, rank)[0] for n in rays)
        assert calls == ([mine] if mine else [])
        assert len(res) == len(rays)
        for s, n in enumerate(rays):
            ref = fake_simulate(np.linspace(0, 359, n), np.full(n, 0.5 + 1.5 * s))
            for k in FIELDS:
                assert res[s][k].shape == (n, N_GATES) and res[s][k].dtype == ref[k].dtype
                assert np.array_equal(res[s][k].view(np.uint8), ref[k].view(np.uint8)), (rank, s, k)


def test_volume_rows_assembled_by_row_gather_equal_the_host_concatenation():
    """VolumeLayout.assemble_on_device (what RadarOperator(distributed=True) runs on the GPU behind the
    all-gather: one row gather per field into the scan-ordered volume, then ONE copy and views) against
    VolumeLayout.assemble (host concatenation, the gloo form) on CPU tensors: mixed dtypes, uneven and
    empty sweeps, more ranks than rays."""
    import torch
    from cosmo_pol_amd.distributed import VolumeLayout
    fields = [('ZH', np.float32), ('lats', np.float64), ('mask', np.float64), ('KDP', np.float32)]
    rng = np.random.default_rng(5)
    for world in (1, 2, 3, 8):
        for n_rays in ([7, 5, 9], [1, 1], [16, 0, 3], [45] * 5):
            ng = 11
            lay = VolumeLayout(fields, n_rays, world, ng)
            truth = [{k: rng.standard_normal((n, ng)).astype(dt) for k, dt in fields} for n in n_rays]
            blocks = np.zeros((world, lay.block.nbytes), dtype=np.uint8)
            for r in range(world):
                row = 0
                for s, b in enumerate(lay.bounds):
                    lo, hi = b[r]
                    for k, _ in lay.block.fields:
                        lay.block.view(blocks[r], k)[row:row + hi - lo] = truth[s][k][lo:hi]
                    row += hi - lo
            rk, rw = lay.source_rows()
            assert len(rk) == len(rw) == sum(n_rays)
            final = torch.zeros(lay.final.nbytes, dtype=torch.uint8)
            lay.assemble_on_device(torch.from_numpy(blocks.reshape(-1)), final)
            got = lay.views_of_final(final.numpy())
            ref = lay.assemble(blocks.reshape(-1))
            for s in range(len(n_rays)):
                for k, dt in fields:
                    assert got[s][k].dtype == dt and got[s][k].shape == (n_rays[s], ng)
                    assert np.array_equal(got[s][k], truth[s][k]) and np.array_equal(ref[s][k], truth[s][k])
                    assert got[s][k].base is not None                  # a view of the one block, not a copy


def _worker_runner(rank, world, port, rays_per_sweep, gather_to, q):
    """ShardedVolumeRunner on CPU tensors (gloo): three scans submitted before the first wait."""
    import ctypes
    sys.path.insert(0, ROOT)
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from cosmo_pol_amd import distributed as D
    fields = list(zip(FIELDS, DTYPES))
    runner = D.ShardedVolumeRunner('cpu', gather_to=gather_to, slots=2)
    calls = []

    def make_run(shift):
        def run_block(a, e, ptrs):              # the stand-in of the kernels: writes through the block's addresses
            calls.append(len(a))
            res = fake_simulate(a, e + shift)
            for k, dt in fields:
                n = len(a) * N_GATES
                buf = (ctypes.c_char * (n * np.dtype(dt).itemsize)).from_address(ptrs[k])
                np.frombuffer(buf, dtype=dt)[:] = res[k].ravel()
        return run_block
    scans = []
    for shift in (0.0, 1.0, 2.0):               # (2 slots, 3 scans: the first slot is used twice)
        sweeps = [(np.linspace(0, 359, n), np.full(n, 0.5 + 1.5 * s)) for s, n in enumerate(rays_per_sweep)]
        scans.append(runner.submit(make_run(shift), 0, sweeps, fields, N_GATES))
    out = []
    for pend in scans:
        res = pend.wait()
        out.append(None if res is None else [{k: v.copy() for k, v in r.items()} for r in res])
    q.put((rank, out, calls, runner.n_collectives, runner.collective))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize('world,rays,gather_to', [(2, (9, 4), None), (2, (9, 4), 0), (3, (10, 7, 1), 2), (2, (1,), 1)])
def test_runner_pending_scans_all_gather_and_rooted_gather(world, rays, gather_to):
    """The product's distributed path (RadarOperator.submit_volume -> ShardedVolumeRunner.submit) with a CPU
    stand-in for the kernels: scans queued back to back, one collective each, results bitwise equal to the
    un-sharded sweeps on the receiving rank(s); with `gather_to` the other ranks get None."""
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_runner, args=(r, world, port, rays, gather_to, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, out, calls, n_coll, name in got:
        assert n_coll == 3 and len(out) == 3
        assert name.startswith('gather(dst=%d)' % gather_to if gather_to is not None else 'all_gather_into_tensor')
        for shift, res in zip((0.0, 1.0, 2.0), out):
            if gather_to is not None and rank != gather_to:
                assert res is None
                continue
            for s, n in enumerate(rays):
                ref = fake_simulate(np.linspace(0, 359, n), np.full(n, 0.5 + 1.5 * s) + shift)
                for k in FIELDS:
                    assert res[s][k].shape == (n, N_GATES) and res[s][k].dtype == ref[k].dtype
                    assert np.array_equal(res[s][k].view(np.uint8), ref[k].view(np.uint8)), (rank, s, k)


def _worker_subgroup(rank, world, port, rays_per_sweep, q):
    """A runner on a SUBGROUP (ranks 1 and 2 of three; root = rank 0 of the subgroup = global rank 1) beside one on the
    default group: what bench.py's single-GPU reference of the c4 speedups does with `dist.new_group([0])`."""
    import ctypes
    sys.path.insert(0, ROOT)
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from cosmo_pol_amd import distributed as D
    fields = list(zip(FIELDS, DTYPES))
    sub = dist.new_group([1, 2])                     # (every rank makes the call)
    sweeps = [(np.linspace(0, 359, n), np.full(n, 0.5 + 1.5 * s)) for s, n in enumerate(rays_per_sweep)]

    def run_block(a, e, ptrs):
        res = fake_simulate(a, e)
        for k, dt in fields:
            n = len(a) * N_GATES
            buf = (ctypes.c_char * (n * np.dtype(dt).itemsize)).from_address(ptrs[k])
            np.frombuffer(buf, dtype=dt)[:] = res[k].ravel()
    out_sub = 'not a member'
    if rank in (1, 2):
        r = D.ShardedVolumeRunner('cpu', group=sub, gather_to=0, slots=2)
        assert (r.world, r.rank) == (2, rank - 1) and r._dst == 1 and r.collective.startswith('gather(dst=0)')
        res = r.submit(run_block, 0, sweeps, fields, N_GATES).wait()
        out_sub = None if res is None else [{k: v.copy() for k, v in x.items()} for x in res]
    full = D.ShardedVolumeRunner('cpu', gather_to=None, slots=2).submit(run_block, 0, sweeps, fields, N_GATES).wait()
    q.put((rank, out_sub, [{k: v.copy() for k, v in x.items()} for x in full]))
    dist.barrier()
    dist.destroy_process_group()


def test_runner_on_a_subgroup_names_its_root_by_global_rank():
    """`dist.gather(dst=...)` takes the root's rank in the DEFAULT group: a runner on a subgroup whose root is not global
    rank 0 must translate (ShardedVolumeRunner._dst).  Ranks 1 and 2 of three shard a scan between them, global rank 1
    receives it; the default-group runner of all three ranks works beside it."""
    world, rays = 3, (9, 4)
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_subgroup, args=(r, world, port, rays, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = {r: (a, b) for r, a, b in (q.get(timeout=120) for _ in range(world))}
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert got[0][0] == 'not a member' and got[2][0] is None and isinstance(got[1][0], list)
    for s, n in enumerate(rays):
        ref = fake_simulate(np.linspace(0, 359, n), np.full(n, 0.5 + 1.5 * s))
        for k in FIELDS:
            assert np.array_equal(got[1][0][s][k].view(np.uint8), ref[k].view(np.uint8)), (s, k)
            for r in range(world):
                assert np.array_equal(got[r][1][s][k].view(np.uint8), ref[k].view(np.uint8)), (r, s, k)
