"""Multi-GPU scans: one process per GPU, rays sharded, ONE collective per scan (volume).

The reference farms the radials of a sweep out to a fork pool and collects
pickled Radial objects over pipes (cosmo_pol/radar_operator.py:402-432).  Here
the rays of a sweep are split into contiguous blocks, one per rank; model cube
and lookup tables are replicated in every GPU's HBM (staged once); every rank
runs its block through the C ABI and the output blocks are collected with a
single all-gather (RCCL over xGMI when the backend is "nccl", gloo on CPU for
tests).  Radials never interact (all coupling - PHIDP / attenuation scans,
sub-beam sums - is inside one radial), so the gathered result is bitwise
identical to the single-GPU result.

Block layout (what one rank contributes, `BlockLayout`): for every output field
a [per, n_gates] array of the field's dtype, the fields one after the other at
8-byte aligned offsets; `per` = ceil(n_rays / world) rows, the unused rows of
the tail ranks stay zero.  The kernels write their outputs straight into the
block on the device (`simulate_sharded_device`), the all-gather runs on device
buffers, and ONE device-to-host copy returns the whole sweep.

A scan of several sweeps (a PPI volume, a set of RHIs) is sharded as a whole
(`VolumeLayout`, `simulate_sharded_volume_device`): a rank takes its contiguous ray block
of EVERY sweep, runs them as one launch sequence (rays of different elevations in one
cpol_run_sweep call), and a single all-gather at the end of the volume assembles all
sweeps on every rank -- 10 kernel launches and one collective per volume instead of 10
launches and one collective per sweep.
"""
import numpy as np


def shard_bounds(n_rays, world, rank):
    """Contiguous block [lo, hi) of rank `rank`; every rank gets ceil(n/world)
    rays except the tail ranks, which may get fewer (or none)."""
    per = -(-int(n_rays) // int(world))
    lo = min(rank * per, n_rays)
    hi = min(lo + per, n_rays)
    return lo, hi, per


def pad_block(arr, per, fill):
    """Pads axis 0 of a rank's block to the common block length `per`."""
    arr = np.asarray(arr)
    if arr.shape[0] == per:
        return arr
    pad = np.full((per - arr.shape[0],) + arr.shape[1:], fill, dtype=arr.dtype)
    return np.concatenate([arr, pad], axis=0)


class BlockLayout(object):
    """Byte layout of one rank's output block: field-major, [per, n_gates] per field."""

    def __init__(self, fields, per, n_gates):
        self.fields = [(k, np.dtype(dt)) for k, dt in fields]
        self.per, self.n_gates = int(per), int(n_gates)
        self.offsets, off = {}, 0
        for k, dt in self.fields:
            off = -(-off // 8) * 8
            self.offsets[k] = off
            off += self.per * self.n_gates * dt.itemsize
        self.nbytes = -(-off // 8) * 8

    def view(self, block, k):
        """[per, n_gates] view of field k inside a uint8 block (numpy)."""
        dt = dict(self.fields)[k]
        o = self.offsets[k]
        return block[o:o + self.per * self.n_gates * dt.itemsize].view(dt).reshape(self.per, self.n_gates)

    def assemble(self, gathered, n_rays, world):
        """[world * nbytes] uint8 (numpy) -> {field: [n_rays, n_gates]} (padding trimmed)."""
        blocks = gathered.reshape(world, self.nbytes)
        out = {}
        for k, dt in self.fields:
            parts = []
            for r in range(world):
                lo, hi, _ = shard_bounds(n_rays, world, r)
                if hi > lo:
                    parts.append(self.view(blocks[r], k)[:hi - lo])
            out[k] = np.concatenate(parts, axis=0) if parts else np.empty((0, self.n_gates), dtype=dt)
        return out


def gather_blocks(local, group=None):
    """All-gathers equally sized per-rank blocks (1-D uint8 torch tensors, CPU or GPU)
    -> [world * nbytes] on every rank.  ONE collective per sweep."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    flat = torch.empty(world * local.numel(), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(flat, local.contiguous().view(-1), group=group)
    return flat


def simulate_sharded(simulate, azimuths, elevations, fields, n_gates, device=None, group=None):
    """Host-buffer form: runs `simulate(az_block, el_block) -> {field: [n_block, n_gates]
    array}` on this rank's contiguous block of rays and gathers all ranks' results.
    `fields`: list of (name, dtype).  Used by the gloo tests with a CPU stand-in and by
    callers whose per-rank results live on the host; the GPU product path is
    `simulate_sharded_device`."""
    import torch
    import torch.distributed as dist
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    az = np.asarray(azimuths, dtype=np.float64)
    el = np.asarray(elevations, dtype=np.float64)
    n = len(az)
    lo, hi, per = shard_bounds(n, world, rank)
    lay = BlockLayout(fields, per, n_gates)
    block = np.zeros(lay.nbytes, dtype=np.uint8)
    if hi > lo:
        res = simulate(az[lo:hi], el[lo:hi])
        for k, dt in lay.fields:
            lay.view(block, k)[:hi - lo] = np.ascontiguousarray(res[k], dtype=dt).reshape(hi - lo, n_gates)
    t = torch.from_numpy(block)
    if device is not None:
        t = t.to(device)
    return lay.assemble(gather_blocks(t, group).cpu().numpy(), n, world)


def simulate_sharded_device(run_block, stream_ptr, azimuths, elevations, fields, n_gates, device,
                            group=None, cache=None):
    """Device-resident form (the product path of RadarOperator(distributed=True)):

      run_block(az_block, el_block, {field: device pointer})   queues the sweep of this
          rank's rays on the library stream `stream_ptr`; the kernels write every output
          field straight into this rank's block of the gather buffer layout;
      one all_gather_into_tensor of the device blocks (RCCL over xGMI), ordered behind the
          kernels with an event;
      one device-to-host copy of the gathered sweep.

    `cache`: dict owned by the caller that keeps the device block / gather buffers between
    sweeps of the same shape."""
    import torch
    import torch.distributed as dist
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    az = np.asarray(azimuths, dtype=np.float64)
    el = np.asarray(elevations, dtype=np.float64)
    n = len(az)
    lo, hi, per = shard_bounds(n, world, rank)
    lay = BlockLayout(fields, per, n_gates)
    cache = {} if cache is None else cache
    key = ('blk', lay.nbytes, world)
    if key not in cache:
        cache.clear()
        cache[key] = (torch.zeros(lay.nbytes, dtype=torch.uint8, device=device),
                      torch.empty(world * lay.nbytes, dtype=torch.uint8, device=device),
                      torch.empty(world * lay.nbytes, dtype=torch.uint8).pin_memory()
                      if torch.cuda.is_available() else torch.empty(world * lay.nbytes, dtype=torch.uint8))
    block, gathered, host = cache[key]
    lib_stream = torch.cuda.ExternalStream(stream_ptr, device=device)
    cur = torch.cuda.current_stream(device)
    lib_stream.wait_stream(cur)                 # the previous gather has consumed the block
    if hi > lo:
        base = block.data_ptr()
        run_block(az[lo:hi], el[lo:hi], {k: base + lay.offsets[k] for k, _ in lay.fields})
    cur.wait_stream(lib_stream)                 # gather behind the kernels
    dist.all_gather_into_tensor(gathered, block, group=group)
    host.copy_(gathered, non_blocking=True)
    cur.synchronize()
    return lay.assemble(host.numpy(), n, world)


class VolumeLayout(object):
    """Sharding of a scan of several sweeps: rank r owns rays [lo, hi) of every sweep
    (`shard_bounds` per sweep) and stores them sweep after sweep in ONE block of
    `BlockLayout(fields, per, n_gates)` rows, per = the largest row count of a rank."""

    def __init__(self, fields, rays_per_sweep, world, n_gates):
        self.world = int(world)
        self.n_rays = [int(n) for n in rays_per_sweep]
        self.bounds = [[shard_bounds(n, self.world, r)[:2] for r in range(self.world)] for n in self.n_rays]
        self.rows = [sum(hi - lo for lo, hi in (b[r] for b in self.bounds)) for r in range(self.world)]
        self.per = max(1, max(self.rows))
        self.block = BlockLayout(fields, self.per, n_gates)
        self.n_gates = int(n_gates)

    def local_rays(self, rank, sweeps):
        """Concatenated (azimuths, elevations) of `rank`: its block of sweep 0, of sweep 1, ..."""
        az = [np.asarray(a, dtype=np.float64)[lo:hi] for (a, _), (lo, hi) in
              zip(sweeps, (b[rank] for b in self.bounds))]
        el = [np.asarray(e, dtype=np.float64)[lo:hi] for (_, e), (lo, hi) in
              zip(sweeps, (b[rank] for b in self.bounds))]
        return (np.concatenate(az) if az else np.empty(0)), (np.concatenate(el) if el else np.empty(0))

    def assemble(self, gathered):
        """[world * nbytes] uint8 (numpy) -> [{field: [n_rays_s, n_gates]} for every sweep s]."""
        blocks = gathered.reshape(self.world, self.block.nbytes)
        out = [dict() for _ in self.n_rays]
        for k, dt in self.block.fields:
            views = [self.block.view(blocks[r], k) for r in range(self.world)]
            row = [0] * self.world
            for s, b in enumerate(self.bounds):
                parts = []
                for r, (lo, hi) in enumerate(b):
                    if hi > lo:
                        parts.append(views[r][row[r]:row[r] + hi - lo])
                        row[r] += hi - lo
                out[s][k] = (np.concatenate(parts, axis=0) if parts
                             else np.empty((0, self.n_gates), dtype=dt))
        return out


def simulate_sharded_volume(simulate, sweeps, fields, n_gates, device=None, group=None):
    """Host-buffer form of the volume sharding (gloo tests, CPU stand-ins): `simulate(az, el) ->
    {field: [n, n_gates]}` is called ONCE with this rank's rays of all sweeps; ONE all-gather."""
    import torch
    import torch.distributed as dist
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    lay = VolumeLayout(fields, [len(a) for a, _ in sweeps], world, n_gates)
    block = np.zeros(lay.block.nbytes, dtype=np.uint8)
    az, el = lay.local_rays(rank, sweeps)
    if len(az):
        res = simulate(az, el)
        for k, dt in lay.block.fields:
            lay.block.view(block, k)[:len(az)] = np.ascontiguousarray(res[k], dtype=dt).reshape(len(az), n_gates)
    t = torch.from_numpy(block)
    if device is not None:
        t = t.to(device)
    return lay.assemble(gather_blocks(t, group).cpu().numpy())


def simulate_sharded_volume_device(run_block, stream_ptr, sweeps, fields, n_gates, device, group=None,
                                   cache=None):
    """Device-resident volume sharding (the product path of RadarOperator(distributed=True)):
    `run_block(az, el, {field: device pointer})` queues ONE launch sequence for this rank's rays of
    all sweeps on the library stream; one all_gather_into_tensor of the device blocks (RCCL over
    xGMI) behind it; one device-to-host copy of the gathered volume.  -> list of per-sweep dicts."""
    import torch
    import torch.distributed as dist
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    lay = VolumeLayout(fields, [len(a) for a, _ in sweeps], world, n_gates)
    nb = lay.block.nbytes
    cache = {} if cache is None else cache
    key = ('blk', nb, world)
    if key not in cache:
        cache.clear()
        cache[key] = (torch.zeros(nb, dtype=torch.uint8, device=device),
                      torch.empty(world * nb, dtype=torch.uint8, device=device),
                      torch.empty(world * nb, dtype=torch.uint8).pin_memory()
                      if torch.cuda.is_available() else torch.empty(world * nb, dtype=torch.uint8))
    block, gathered, host = cache[key]
    lib_stream = torch.cuda.ExternalStream(stream_ptr, device=device)
    cur = torch.cuda.current_stream(device)
    lib_stream.wait_stream(cur)                 # the previous gather has consumed the block
    az, el = lay.local_rays(rank, sweeps)
    if len(az):
        base = block.data_ptr()
        run_block(az, el, {k: base + lay.block.offsets[k] for k, _ in lay.block.fields})
    cur.wait_stream(lib_stream)                 # gather behind the kernels
    dist.all_gather_into_tensor(gathered, block, group=group)
    host.copy_(gathered, non_blocking=True)
    cur.synchronize()
    return lay.assemble(host.numpy())
